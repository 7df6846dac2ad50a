"""Phase timeline of the implicit-GEMM convolution kernel from in-kernel clock stamps (diagnostic build libconvdbg.so:
hipcc -DCONV_STAMPS ... all csrc/*.hip).  python tools/micro/dbg_conv.py"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import dm_amd_bootstrap  # noqa
from dm_amd.runtime import hip, ops
hip.LIB_PATH = os.path.join(ROOT, "tools/micro/libconvdbg.so")
hip._lib = None
lib = hip.lib()
lib.ddpm_conv_set_dbg.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0"); BF = torch.bfloat16; N = 256
names = ["pixel table", "stage chunk 0", "mfma chunk 0", "other chunks", "sums", "acc->LDS", "store"]
for (hw, cin, cout) in [(32, 32, 32), (16, 32, 32), (8, 64, 64), (8, 128, 64)]:
    x = torch.randn(N, hw, hw, cin, device=dev).to(BF)
    w = (torch.randn(9, cout, cin, device=dev) * 0.05).to(BF)
    b = torch.zeros(cout, device=dev)
    run = lambda: ops.conv_forward(0, x, w, b, cout)
    run(); torch.cuda.synchronize()
    dbg = torch.zeros(8192 * 4 * 16, device=dev, dtype=torch.int64)
    lib.ddpm_conv_set_dbg(dbg.data_ptr())
    run(); torch.cuda.synchronize()
    lib.ddpm_conv_set_dbg(None)
    d = dbg.view(-1, 16).cpu()
    d = d[d[:, 0] > 0][:, :8].double()
    ph = (d[:, 1:] - d[:, :-1]).mean(0)
    start = d[:, 0] - d[:, 0].min()
    print(f"{hw}x{hw} {cin}->{cout}: waves {len(d)}; wave lifetime {(d[:, 7] - d[:, 0]).mean():.0f} ticks; starts spread {start.max():.0f}; last end {(d[:, 7].max() - d[:, 0].min()):.0f}")
    print("   " + "  ".join(f"{n} {v:.0f}" for n, v in zip(names, ph.tolist())))
