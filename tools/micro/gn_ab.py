"""GroupNorm + activation forward / backward timing (graph of 20 launches) for large-image shapes; optional argument: another
build of the library to time instead (python tools/micro/gn_ab.py [path/to/libddpm_hip.so])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import dm_amd_bootstrap  # noqa
from dm_amd.runtime import hip, ops
if len(sys.argv) > 1:
    hip.LIB_PATH = os.path.abspath(sys.argv[1]); hip._lib = None
    import ctypes
    _probe = ctypes.CDLL(hip.LIB_PATH)
    hip._PROTOS = {k: v for k, v in hip._PROTOS.items() if hasattr(_probe, k)}     # an older build lacks the newer entry points
dev = torch.device("cuda:0"); BF = torch.bfloat16


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / (5 * reps)


for (n, h, c) in [(32, 128, 32), (32, 128, 64), (32, 64, 64), (32, 64, 128), (128, 64, 32), (256, 16, 32)]:
    x = torch.randn(n, h, h, c, device=dev).to(BF)
    dy = torch.randn_like(x)
    gamma, beta = torch.rand(c, device=dev) + 0.5, torch.rand(c, device=dev)
    y, st = ops.gn_act_forward(x, gamma, beta, 16, hip.ACT["gelu"])
    part = torch.zeros(n, 3, c, device=dev)
    out, dx = torch.empty_like(x), torch.empty_like(x)
    f = timed(lambda: ops.gn_act_forward(x, gamma, beta, 16, hip.ACT["gelu"], out=out, stats=st))
    b = timed(lambda: ops.gn_act_backward(x, dy, gamma, beta, 16, hip.ACT["gelu"], st, part, out=dx))
    gb = x.numel() * 2 / 1e9
    print(f"N{n} {h}x{h} C{c}: fwd {f:7.1f} us ({2 * gb / f * 1e6 / 1e3:5.2f} TB/s)   bwd {b:7.1f} us ({3 * gb / b * 1e6 / 1e3:5.2f} TB/s)")
