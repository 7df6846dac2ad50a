# Runs on the GPU box: rocprofv3 kernel statistics of 32x32x1 sampling steps (one chain of 512 images), top kernels by time.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p32 && mkdir -p /tmp/p32
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p32 -o r -- python3 $R/tools/micro/sample32.py ${1:-512} > /tmp/p32/o.txt 2> /tmp/p32/e.txt
cat /tmp/p32/o.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/p32/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:30]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
