"""The depthwise 3x3 weight gradient timed alone (HIP events, 20 launches): python tools/micro/dw_wgrad_time.py [batch]
DDPM_DW_WGRAD_CAP=n in the environment bounds the workgroups of a launch (a measurement switch of the library)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402,F401  (puts the package on the path under its import name)
from dm_amd.runtime import ops  # noqa: E402

dev = torch.device("cuda", 0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for h, c in ((64, 64), (128, 64), (32, 128)):
    x = torch.randn(nb, h, h, c, device=dev).bfloat16()
    dy = torch.randn_like(x)
    blocks = ops.dwconv3x3_wgrad_blocks(nb, h, h, c)
    part = torch.empty(blocks * 10 * c, device=dev)
    for _ in range(3):
        ops.dwconv3x3_wgrad(x, dy, part)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        ops.dwconv3x3_wgrad(x, dy, part)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print(f"{h}x{h} C{c} N{nb}: {blocks} workgroups {us:7.1f} us  {2 * x.numel() * 2 / us / 1e3:7.0f} GB/s", flush=True)
