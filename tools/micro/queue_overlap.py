"""From a rocprofv3 --kernel-trace CSV: over the last `window_ms` of the trace, per-queue busy time, and how long 1, 2, 3 ...
queues were busy at the same time.   python tools/micro/queue_overlap.py <kernel_trace.csv> [window_ms]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 20e6
end = max(int(r["End_Timestamp"]) for r in rows)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= end - win]
byq = collections.defaultdict(float)
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    byq[r["Queue_Id"]] += e - s
    ev.append((s, 1)); ev.append((e, -1))
print("window ms:", win / 1e6, "kernels:", len(rows))
for q, b in sorted(byq.items()):
    print(f"  queue {q}: busy {b / 1e6:.3f} ms")
ev.sort()
depth, last, hist = 0, ev[0][0], collections.defaultdict(float)
for t, d in ev:
    hist[depth] += t - last
    last = t
    depth += d
for k in sorted(hist):
    print(f"  {k} kernels in flight: {hist[k] / 1e6:.3f} ms")
# busy segments per queue (kernels less than 30 us apart merged), in time order, over the last 12 ms
segs = []
cur = {}
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    if s < end - 12e6:
        continue
    if q in cur and s - cur[q][1] < 30e3:
        cur[q][1] = max(cur[q][1], e); cur[q][2] += 1
    else:
        if q in cur:
            segs.append((cur[q][0], cur[q][1], q, cur[q][2]))
        cur[q] = [s, e, 1]
for q, c in cur.items():
    segs.append((c[0], c[1], q, c[2]))
for s, e, q, n in sorted(segs):
    print(f"  q{q}: {(s - (end - 12e6)) / 1e3:9.1f} .. {(e - (end - 12e6)) / 1e3:9.1f} us  ({n} kernels)")
