# Runs on the GPU box: the kernels of ONE pipelined training step (default bench configuration) in start order, both queues,
# from a rocprofv3 trace: start (us), duration, queue, grid, name.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pts && mkdir -p /tmp/pts
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pts -o r -- python3 $R/bench.py --steps 8 --warmup 3 --no-sample --no-cpu --no-roofline --no-extra $BENCH_ARGS > /tmp/pts/o.txt 2> /tmp/pts/e.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/pts/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('bump_step_kernel')]      # one per step, right after its last Adam launch
a, b = idx[-3] + 1, idx[-2] + 1
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    g = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0)
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:6.1f}  q{r.get('Queue_Id', '?'):>2} grid {g:8d} {r['Kernel_Name'][:70]}")
PY
