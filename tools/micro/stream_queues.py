"""Which hardware queue does the k-th stream a process creates land on?  Every stream runs one elementwise kernel whose grid
size identifies it; read the queue ids from a rocprofv3 --kernel-trace of this script (tools/micro/stream_queues.py report <csv>)."""
import csv
import sys

if len(sys.argv) > 2 and sys.argv[1] == "report":
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "elementwise" in r["Kernel_Name"]]
    for r in sorted(rows, key=lambda r: int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))):
        print("grid", r.get("Grid_Size_X", r.get("Grid_Size")), "queue", r["Queue_Id"])
    sys.exit(0)
import torch

dev = torch.device("cuda:0")
xs = [torch.zeros(256 * 256 * (k + 1), device=dev) for k in range(12)]
torch.cuda.synchronize()
xs[0].add_(1)                       # the default stream
prio = {8: -1, 9: -1, 10: 0, 11: 0}
for k in range(1, 12):
    s = torch.cuda.Stream(priority=prio.get(k, 0))
    with torch.cuda.stream(s):
        xs[k].add_(1)
torch.cuda.synchronize()
