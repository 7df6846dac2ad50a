"""Phase timeline of the fused inverted-residual BACKWARD kernel from in-kernel clock stamps (diagnostic build tools/micro/lib_sirdbg.so:
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSIR_STAMPS -I diffusion-model_amd/csrc -I include diffusion-model_amd/csrc/*.hip).  python tools/micro/dbg_sir_bwd.py"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import dm_amd_bootstrap  # noqa
from dm_amd.runtime import hip, ops
hip.LIB_PATH = os.path.join(ROOT, "tools/micro/lib_sirdbg.so")
hip._lib = None
lib = hip.lib()
lib.ddpm_sir_bwd_set_dbg.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0"); BF = torch.bfloat16; N = 256
names = ["loads+tables", "GN3 bwd", "proj dgrad GEMM", "GN2 bwd", "depthwise", "GN1 bwd+store", "expand dgrad GEMM", "dx store"]
names_big = ["GN3 bwd", "proj dgrad GEMM", "GN2 bwd", "depthwise", "tap sums", "GN1 bwd+store", "expand dgrad GEMM", "dx store"]
for (c, w) in [(32, 32), (32, 16), (64, 8), (128, 4), (256, 2)]:
    e = 2 * c
    x = torch.randn(N, w, w, c, device=dev).to(BF)
    f = lambda *s: torch.randn(*s, device=dev) * 0.1
    we, wp = (torch.randn(1, e, c, device=dev) * 0.1).to(BF), (torch.randn(1, c, e, device=dev) * 0.1).to(BF)
    g1, b1, g2, b2, g3 = 1 + f(e), f(e), 1 + f(e), f(e), 1 + f(c)
    masks = tuple(torch.randint(0, 256, (N * w * w * ch // 8,), device=dev, dtype=torch.uint8) for ch in (c, e, e))
    y, sv = ops.sir_forward(x, we, f(e), g1, b1, f(9, e), f(e), g2, b2, wp, f(c), g3, f(c), 16, hip.ACT["gelu"], masks=masks, mask_scale=1.1, record=True)
    wpd, wed = (torch.randn(1, e, c, device=dev) * 0.1).to(BF), (torch.randn(1, c, e, device=dev) * 0.1).to(BF)
    dy = torch.randn_like(x)
    nwg = ops.sir_workgroups(w * w, N)
    p1, p2, p3 = (torch.zeros(N, 3, ch, device=dev) for ch in (e, e, c))
    dwp = torch.zeros(nwg, 10, e, device=dev)
    run = lambda: ops.sir_backward(dy, sv, g1, b1, g2, b2, g3, wpd, wed, f(9, e), 16, hip.ACT["gelu"], p1, p2, p3, dwp, masks=masks, mask_scale=1.1)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    dbg = torch.zeros(N * 8 * 16, device=dev, dtype=torch.int64)
    lib.ddpm_sir_bwd_set_dbg(dbg.data_ptr())
    run(); torch.cuda.synchronize()
    lib.ddpm_sir_bwd_set_dbg(None)
    d = dbg.view(-1, 16).cpu()
    d = d[d[:, 0] > 0][:, :9].double()
    ph = (d[:, 1:] - d[:, :-1]).mean(0)
    tot = (d[:, 8] - d[:, 0]).mean().item()
    span = d[:, 8].max().item() - d[:, 0][d[:, 0] > d[:, 0].median() - 1e7].min().item()
    print(f"C{c} {w}x{w}: waves {len(d)}; wave lifetime {tot:.0f} ticks; first start -> last end {span:.0f} ticks; kernel {us:.1f} us (with allocs/launch)")
    print("   " + "  ".join(f"{n} {v:.0f}" for n, v in zip(names_big if w == 32 else names, ph.tolist())))
