"""Every launch of one denoiser forward on 3-channel images (128x128 with self-attention at the 16x16 stage by default), timed
alone with HIP events, largest first:
python tools/micro/sample128_launches.py [batch [size]]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import build_model  # noqa: E402
from dm_amd.runtime import ops, roofline  # noqa: E402

dev = torch.device("cuda", 0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
m = build_model(dev, 3, [2] if size == 128 else [])
m.eval()
x = torch.randn(nb, 3, size, size, device=dev)
t = torch.randint(0, 120, (nb,), device=dev)
with torch.no_grad():
    m(x[:nb // 2], t[:nb // 2])          # another shape first (one-time set-up); the module replays a graph from a shape's second pass on
    probe = roofline._Probe()
    ops.PROBE = probe
    try:
        m(x, t)
    finally:
        ops.PROBE = None
torch.cuda.synchronize()
rows = [(e0.elapsed_time(e1) * 1e3, k, tag, nbytes) for k, e0, e1, nbytes, flops, tag in probe.rows]
tot = sum(r[0] for r in rows)
print(f"{len(rows)} launches, {tot / 1e3:.2f} ms")
for us, k, tag, nbytes in sorted(rows, reverse=True)[:45]:
    print(f"{us:8.1f} us  {k:28s} {tag:40s} {nbytes / us * 1e-3:7.0f} GB/s")
