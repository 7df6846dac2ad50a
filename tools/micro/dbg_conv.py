import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import dm_amd_bootstrap  # noqa
from dm_amd.runtime import hip, ops, pack
hip.LIB_PATH = os.path.join(ROOT, "tools/micro/libdbg.so")
hip._lib = None
lib = hip.lib()
lib.ddpm_set_dbg.argtypes = [ctypes.c_uint64]
dev = torch.device("cuda:0"); BF = torch.bfloat16; N = 256
dbg = torch.zeros(8 * 65536, device=dev, dtype=torch.int64)
for (kind, h, ci, co) in [(hip.CONV_1X1, 1, 32, 64), (hip.CONV_3X3_S1, 16, 32, 32), (hip.CONV_1X1, 16, 32, 64), (hip.CONV_3X3_S1, 2, 256, 256), (hip.CONV_3X3_S1, 32, 32, 32)]:
    k = {hip.CONV_3X3_S1: 3, hip.CONV_1X1: 1}[kind]
    x = torch.randn(N, h, h, ci, device=dev).to(BF)
    w = pack.pack_fwd(kind, torch.randn(co, ci, k, k, device=dev) * 0.05); b = torch.zeros(co, device=dev)
    out = ops.conv_forward(kind, x, w, b, co)
    for rep in range(3):
        dbg.zero_(); torch.cuda.synchronize()
        lib.ddpm_set_dbg(dbg.data_ptr())
        ops.conv_forward(kind, x, w, b, co, out=out); torch.cuda.synchronize()
        lib.ddpm_set_dbg(0)
    d = dbg.view(-1, 8).cpu()
    d = d[d[:, 0] > 0]
    t0 = d[:, 0].min().item()
    ph = (d[:, 1:6] - d[:, 0:5]).float().mean(0) * 10  # ns (100 MHz ticks)
    print(f"kind {kind} [{N},{h},{h},{ci}]->{co}: blocks {len(d)}; per-block ns: setup {ph[0]:.0f} stage {ph[1]:.0f} sync {ph[2]:.0f} mfma(all chunks) {ph[3]:.0f} epilogue {ph[4]:.0f}; "
          f"block lifetime avg {(d[:,5]-d[:,0]).float().mean().item()*10:.0f} ns; first start -> last end {(d[:,5].max().item()-t0)*10} ns; last block start {(d[:,0].max().item()-t0)*10} ns")
