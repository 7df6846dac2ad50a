# kernel + memory-copy trace of a few training steps; prints every activity of the window around one step boundary
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof/t -o r -- python3 $R/bench.py --steps 12 --warmup 5 --no-sample --no-cpu --no-roofline --no-extra > /tmp/prof/o.json 2> /tmp/prof/err.txt
ls /tmp/prof/t/*/ | head
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob('/tmp/prof/t/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K q%s %s' % (r['Queue_Id'], r['Kernel_Name'][:60])))
for f in glob.glob('/tmp/prof/t/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'M %s %s bytes' % (r.get('Direction', ''), r.get('Bytes', r.get('Size', '')))))
ev.sort()
# last adam kernel but three -> print 400 us after it
adams = [i for i, e in enumerate(ev) if 'adam_kernel' in e[2]]
i0 = adams[-3]
t0 = ev[i0][0]
for s, e, name in ev[i0 - 6:]:
    if s - t0 > 450e3:
        break
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f}  {name}")
PY
