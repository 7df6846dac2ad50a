"""Phase timeline of the fused inverted-residual kernel from in-kernel clock stamps (diagnostic build libsirdbg.so)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import dm_amd_bootstrap  # noqa
from dm_amd.runtime import hip, ops, pack
hip.LIB_PATH = os.path.join(ROOT, "tools/micro/libsirdbg.so")
hip._lib = None
lib = hip.lib()
lib.ddpm_sir_set_dbg.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0"); BF = torch.bfloat16; N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
names = ["prefetch+X->LDS", "expand GEMM", "read tile", "stats1", "apply1", "depthwise", "stats2", "apply2", "project GEMM", "GN3+res"]
for (c, w) in [(32, 16), (64, 8), (128, 4), (256, 2)]:
    e = 2 * c
    x = torch.randn(N, w, w, c, device=dev).to(BF)
    f = lambda *s: torch.randn(*s, device=dev) * 0.1
    we, wp = (torch.randn(1, e, c, device=dev) * 0.1).to(BF), (torch.randn(1, c, e, device=dev) * 0.1).to(BF)
    args = (x, we, f(e), 1 + f(e), f(e), f(9, e), f(e), 1 + f(e), f(e), wp, f(c), 1 + f(c), f(c), 16, hip.ACT["gelu"])
    masks = tuple(torch.randint(0, 256, (N * w * w * ch // 8,), device=dev, dtype=torch.uint8) for ch in (c, e, e))
    for rec in (False, True):
        dbg = torch.zeros(N * 8 * 16, device=dev, dtype=torch.int64)
        kw = dict(masks=masks if rec else (None, None, None), mask_scale=1.1, record=rec)
        ops.sir_forward(*args, **kw); torch.cuda.synchronize()
        lib.ddpm_sir_set_dbg(dbg.data_ptr())
        ops.sir_forward(*args, **kw); torch.cuda.synchronize()
        lib.ddpm_sir_set_dbg(None)
        d = dbg.view(-1, 16).cpu()
        d = d[d[:, 0] > 0][:, :10].double()
        ph = (d[:, 1:] - d[:, :-1]).mean(0)
        tot = (d[:, 9] - d[:, 0]).mean().item()
        span = d[:, 9].max().item() - d[:, 0].min().item()
        print(f"C{c} {w}x{w} rec={rec}: waves {len(d)}; wave lifetime {tot:.0f} ticks; first start -> last end {span:.0f} ticks")
        print("   " + "  ".join(f"{n} {v:.0f}" for n, v in zip(names[:9], ph.tolist())))
        f = dbg.view(-1, 16).cpu()
        f = f[f[:, 0] > 0].double()
        if (f[:, 13] > 0).all():           # finer stamps of the second statistics phase (the last call overwrites the first)
            base = f[:, 5]
            print("   stats2 detail: accumulate %.0f  shuffle+store %.0f  barrier %.0f  finish %.0f  barrier %.0f" % tuple(
                (f[:, i] - f[:, j]).mean().item() for i, j in ((10, 5), (11, 10), (12, 11), (13, 12), (6, 13))))
