"""Times conv -> GroupNorm units (fused whole-image launch vs convolution + GroupNorm launches) at the benchmark's shapes.
Usage: python tools/micro/bench_img.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dm_amd_bootstrap  # noqa: F401,E402
from dm_amd.runtime import hip, ops, pack  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""
K33, K33S2, KT = hip.CONV_3X3_S1, hip.CONV_3X3_S2, hip.CONVT_2X2_S2
S = int(sys.argv[3]) if len(sys.argv) > 3 else 32       # input image size of the network: stage resolutions scale with it
FWD = [("a s2 32->32 32^2", K33S2, 32, 32, 32), ("b 3x3 32->32 16^2", K33, 32, 32, 16), ("c s2 32->64 16^2", K33S2, 32, 64, 16),
       ("d s2 64->128 8^2", K33S2, 64, 128, 8), ("e s2 128->256 4^2", K33S2, 128, 256, 4), ("f convT 256->128 2^2", KT, 256, 128, 2),
       ("g convT 128->64 4^2", KT, 128, 64, 4), ("h convT 64->32 8^2", KT, 64, 32, 8), ("j convT 32->32 16^2", KT, 32, 32, 16)]


def timeit(fn, reps=20):
    """device time per call: the calls are captured into one graph (no host launch cost between them) and replayed"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (5 * reps)


for tag, kind, ci, co, h in FWD:
    if ONLY and ONLY != "-" and not tag.startswith(ONLY):
        continue
    h = h * S // 32
    tag = f"{tag.split()[0]} {tag.split()[1]} {tag.split()[2]} {h}^2"
    k = 2 if kind == KT else 3
    x = torch.randn(N, h, h, ci, device=dev).bfloat16()
    w = torch.randn((ci, co, k, k) if kind == KT else (co, ci, k, k), device=dev) * 0.05
    wf, wd = pack.pack_fwd(kind, w), pack.pack_dgrad(kind, w)
    b = torch.randn(co, device=dev) * 0.1
    gamma, beta = torch.ones(co, device=dev), torch.zeros(co, device=dev)
    act = hip.ACT["gelu"]
    fused = lambda: ops.conv_forward(kind, x, wf, b, co, gn=(gamma, beta, 16, act, None))           # noqa: E731

    def split():
        raw = ops.conv_forward(kind, x, wf, b, co)
        return ops.gn_act_forward(raw, gamma, beta, 16, act)

    ok = ops.conv_gn_fusable(kind, x, None, co, 16)
    t_f = timeit(fused) if ok else float("nan")
    t_s = timeit(split)
    raw, y, stats = fused() if ok else (split()[0],) * 3
    oh = raw.shape[1]
    dy = torch.randn(N, oh, oh, co, device=dev).bfloat16()
    raw = ops.conv_forward(kind, x, wf, b, co)
    _, stats = ops.gn_act_forward(raw, gamma, beta, 16, act)
    dk = pack.DGRAD_KIND[kind]
    part = torch.zeros(N, 3, co, device=dev)
    draw = torch.empty_like(raw)
    bfused = lambda: ops.conv_forward(dk, dy, wd, None, ci, rows=ci, gnb=(raw, gamma, beta, 16, act, stats, part, draw))   # noqa: E731

    def bsplit():
        d_ = ops.gn_act_backward(raw, dy, gamma, beta, 16, act, stats, part)
        return ops.conv_forward(dk, d_, wd, None, ci, rows=ci)

    okb = ops.conv_gnb_fusable(dk, dy, ci, 16)
    t_bf = timeit(bfused) if okb else float("nan")
    t_bs = timeit(bsplit)
    print(f"{tag:24s} fwd fused {t_f:7.1f} us  split {t_s:7.1f} us | bwd fused {t_bf:7.1f} us  split {t_bs:7.1f} us", flush=True)
