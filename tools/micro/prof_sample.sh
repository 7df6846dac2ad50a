set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
for cfg in "512 2" "1024 1" "512 1"; do timeout -k 10 120 python3 $R/tools/micro/sample_only.py $cfg 2>/dev/null; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/s -o r -- python3 $R/tools/micro/sample_only.py 1024 1 > /tmp/prof/o.txt 2> /tmp/prof/err.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof/s/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:28]:
    print(f"{r['Name'][:64]:64s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:7.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
