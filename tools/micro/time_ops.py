"""Isolated timings of the layer kernels on the benchmark's shapes (back-to-back launches, CUDA events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dm_amd_bootstrap  # noqa
from dm_amd.runtime import hip, ops, pack

dev = torch.device("cuda:0")
BF = torch.bfloat16


def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


N = 256
print("-- GroupNorm forward")
for (h, c, g) in [(16, 32, 16), (16, 64, 16), (8, 64, 16), (8, 128, 16), (4, 128, 16), (4, 256, 16), (2, 256, 16), (2, 512, 16), (32, 32, 16), (32, 64, 16)]:
    x = torch.randn(N, h, h, c, device=dev).to(BF)
    gam, bet = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    out = torch.empty_like(x); stats = torch.empty(N, g, 2, device=dev)
    mb = x.numel() * 2 / 1e6
    for label, kw in [("gelu", dict(act=1)), ("none", dict(act=0))]:
        us = timeit(lambda: ops.gn_act_forward(x, gam, bet, g, kw["act"], stats=stats, out=out))
        print(f"  [{N},{h},{h},{c}] {label}: {us:7.1f} us   ({2*mb/us/1e6*1e6/1e6:.2f} TB/s for {2*mb:.1f} MB)")
print("-- conv forward")
for (kind, h, ci, co) in [(hip.CONV_3X3_S1, 16, 32, 32), (hip.CONV_1X1, 16, 32, 64), (hip.CONV_1X1, 16, 64, 32), (hip.CONV_3X3_S2, 16, 32, 64), (hip.CONV_1X1, 8, 64, 128),
                          (hip.CONV_3X3_S1, 8, 64, 64), (hip.CONV_3X3_S1, 4, 128, 128), (hip.CONV_3X3_S1, 2, 256, 256), (hip.CONV_3X3_S1, 2, 256, 512), (hip.CONV_3X3_S1, 2, 512, 256),
                          (hip.CONV_1X1, 2, 256, 512), (hip.CONV_1X1, 1, 32, 64), (hip.CONV_3X3_S1, 32, 32, 32), (hip.CONV_3X3_S1, 32, 16, 32)]:
    k = {hip.CONV_3X3_S1: 3, hip.CONV_3X3_S2: 3, hip.CONV_1X1: 1}[kind]
    x = torch.randn(N, h, h, ci, device=dev).to(BF)
    w = pack.pack_fwd(kind, torch.randn(co, ci, k, k, device=dev) * 0.05)
    b = torch.zeros(co, device=dev)
    out = ops.conv_forward(kind, x, w, b, co)
    us = timeit(lambda: ops.conv_forward(kind, x, w, b, co, out=out))
    flops = 2.0 * out.numel() * ci * k * k
    mb = (x.numel() + out.numel()) * 2 / 1e6
    print(f"  kind {kind} [{N},{h},{h},{ci}]->{co}: {us:7.1f} us   {flops/us/1e6:.1f} TFLOP/s, {mb/us/1e6*1e6/1e6:.2f} TB/s act traffic ({mb:.1f} MB)")
print("-- depthwise forward")
for (h, c) in [(16, 64), (8, 128), (4, 256), (2, 512), (32, 64)]:
    x = torch.randn(N, h, h, c, device=dev).to(BF)
    w9 = torch.randn(9, c, device=dev); b = torch.zeros(c, device=dev); out = torch.empty_like(x)
    us = timeit(lambda: ops.dwconv3x3(x, w9, b, out=out))
    mb = x.numel() * 4 / 1e6
    print(f"  [{N},{h},{h},{c}]: {us:7.1f} us  {mb/us:.2f} TB/s ({mb:.1f} MB)")
