"""Per-launch timing of one inference forward pass (events around every launch): python tools/micro/lds_probe.py [batch] [size]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import build_model  # noqa: E402
from dm_amd.runtime import ops, roofline  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
size = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda", 0)
m = build_model(dev, 1 if size == 32 else 3)
m.eval()
m.model.use_graphs = False
x = torch.randn(nb, 1 if size == 32 else 3, size, size, device=dev)
t = torch.full((nb,), 60, device=dev, dtype=torch.long)
with torch.no_grad():
    for _ in range(3):
        m(x, t)
    probe = roofline._Probe()
    ops.PROBE = probe
    for _ in range(3):
        m(x, t)
    torch.cuda.synchronize()
    ops.PROBE = None
rows = probe.rows[len(probe.rows) * 2 // 3:]
tot = 0.0
for kernel, e0, e1, nbytes, flops, tag in rows:
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    print(f"{kernel:28s} {us:9.1f} us  {nbytes / 1e6:8.2f} MB {flops / 1e9:8.2f} GFLOP  {tag}")
print(f"total {tot:.1f} us for batch {nb}")
