"""The bandwidth-bound per-op kernels of large images, timed alone (HIP events, 20 launches each) with their algorithmic GB/s:
python tools/micro/large_image_ops.py [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402,F401  (puts the package on the path under its import name)
from dm_amd.runtime import hip, ops  # noqa: E402

dev = torch.device("cuda", 0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for h, c in ((64, 64), (128, 64), (128, 32), (64, 128), (128, 64)):
    x = torch.randn(nb, h, h, c, device=dev).bfloat16()
    w9 = torch.randn(9, c, device=dev)
    b = torch.randn(c, device=dev)
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    res = torch.randn_like(x)
    gb = x.numel() * 2 / 1e9
    y = torch.empty_like(x)
    us = timed(lambda: ops.dwconv3x3(x, w9, b, out=y))
    print(f"{h}x{h} C{c} N{nb}: depthwise           {us:7.1f} us  {2 * gb / us * 1e6:7.0f} GB/s", flush=True)
    us = timed(lambda: ops.dwconv3x3(x, w9, b, out=y, gn_part_groups=16))
    print(f"{h}x{h} C{c} N{nb}: depthwise + sums    {us:7.1f} us  {2 * gb / us * 1e6:7.0f} GB/s", flush=True)
    _, part = ops.dwconv3x3(x, w9, b, out=y, gn_part_groups=16)
    us = timed(lambda: ops.gn_act_forward(x, gamma, beta, 16, hip.ACT["gelu"], out=y, stat_part=part))
    print(f"{h}x{h} C{c} N{nb}: GroupNorm apply     {us:7.1f} us  {2 * gb / us * 1e6:7.0f} GB/s", flush=True)
    us = timed(lambda: ops.gn_act_forward(x, gamma, beta, 16, 0, out=y, stat_part=part, res=res))
    print(f"{h}x{h} C{c} N{nb}: GroupNorm apply+res {us:7.1f} us  {3 * gb / us * 1e6:7.0f} GB/s", flush=True)
    us = timed(lambda: ops.gn_act_forward(x, gamma, beta, 16, hip.ACT["gelu"], out=y))
    print(f"{h}x{h} C{c} N{nb}: GroupNorm two-pass  {us:7.1f} us  {3 * gb / us * 1e6:7.0f} GB/s", flush=True)
    us = timed(lambda: y.copy_(x))
    print(f"{h}x{h} C{c} N{nb}: copy                {us:7.1f} us  {2 * gb / us * 1e6:7.0f} GB/s", flush=True)
