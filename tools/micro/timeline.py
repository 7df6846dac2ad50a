"""Summarise a rocprofv3 --kernel-trace CSV: for the last full training step (between two bump_step_kernel launches: one per
step, right after its last Adam launch) print,
per queue, busy time and gaps, and list the kernels of the less busy queue with their offsets.
python tools/micro/timeline.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adams = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("bump_step_kernel")]
a, b = adams[-2], adams[-1]
step = rows[a + 1:b + 1]
t0 = int(rows[a]["End_Timestamp"])
print("step length us:", (int(rows[b]["End_Timestamp"]) - t0) / 1e3, "kernels:", len(step))
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e3
    print(f"queue {q}: {len(rs)} kernels, busy {busy:.1f} us, first at {(int(rs[0]['Start_Timestamp']) - t0) / 1e3:.1f}, last ends {(int(rs[-1]['End_Timestamp']) - t0) / 1e3:.1f}")
main = max(byq.values(), key=len)
gaps = []
for x, y in zip(main, main[1:]):
    g = (int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e3
    gaps.append((g, x["Kernel_Name"][:40], y["Kernel_Name"][:40], (int(x["End_Timestamp"]) - t0) / 1e3))
print("main-queue gap total us:", sum(g for g, *_ in gaps if g > 0), " largest:")
for g in sorted(gaps, reverse=True)[:12]:
    print("  %.1f us after %s before %s at %.1f" % g)
for q, rs in byq.items():
    if rs is main:
        continue
    print("queue", q)
    for r in rs:
        print(f"   {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} .. {(int(r['End_Timestamp']) - t0) / 1e3:8.1f}  {r['Kernel_Name'][:70]}")
fam = collections.defaultdict(lambda: [0, 0.0])
import re
for r in main:
    k = re.split(r"[<(]", r["Kernel_Name"].replace("void ", ""))[0]
    fam[k][0] += 1
    fam[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("main queue by kernel family:")
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"   {k:50s} x{c:3d} {t:8.1f} us")
shape = collections.defaultdict(lambda: [0, 0.0])
for r in main:
    k = r["Kernel_Name"].replace("void ", "")[:48] + " grid " + r.get("Grid_Size_X", r.get("Grid_Size", "?"))
    shape[k][0] += 1
    shape[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("main queue by kernel and grid:")
for k, (c, t) in sorted(shape.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"   {k:72s} x{c:3d} avg {t / c:7.1f} total {t:8.1f} us")
print("main queue, first kernels of the step:")
for r in main[:34]:
    print(f"   {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} .. {(int(r['End_Timestamp']) - t0) / 1e3:8.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8s}  {r['Kernel_Name'][:110]}")
print("main queue, first and last kernels:")
for r in main[:10] + main[-8:]:
    print(f"   {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} .. {(int(r['End_Timestamp']) - t0) / 1e3:8.1f}  {r['Kernel_Name'][:70]}  grid {r.get('Grid_Size', r.get('Grid_Size_X', ''))}")
if len(sys.argv) > 2:            # second argument: file for the whole main-queue sequence
    with open(sys.argv[2], "w") as f:
        for r in main:
            f.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  {r['Kernel_Name'][:90]}\n")
