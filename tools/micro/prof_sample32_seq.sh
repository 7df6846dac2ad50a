# Runs on the GPU box: the kernels of ONE 32x32x1 sampling step (one chain of 512 images) in start order, from a rocprofv3 trace.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p32s && mkdir -p /tmp/p32s
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p32s -o r -- python3 $R/tools/micro/sample32.py ${1:-512} > /tmp/p32s/o.txt 2> /tmp/p32s/e.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/p32s/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the last complete step: between the last two reverse_step kernels
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('reverse_step')]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
for r in rows[a:b + 4]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:6.1f} gap {(s - prev_end) / 1e3:5.1f}  q{r.get('Queue_Id', '?'):>2} {r['Kernel_Name'][:80]}")
    prev_end = max(prev_end, e)
PY
