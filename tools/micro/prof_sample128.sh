set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p128 && mkdir -p /tmp/p128
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p128 -o r -- python3 $R/tools/micro/sample128.py 256 > /tmp/p128/o.txt 2> /tmp/p128/e.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/p128/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:24]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
