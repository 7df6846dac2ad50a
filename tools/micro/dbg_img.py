"""Phase timeline of the whole-image convolution kernel from in-kernel clock stamps (100 MHz constant clock).  Builds a
diagnostic library with -DIMG_STAMPS first:  python tools/micro/dbg_img.py build   (CPU box), then run on the GPU box."""
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "tools/micro/libimgdbg.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    src = sorted(glob.glob(os.path.join(ROOT, "diffusion-model_amd/csrc/*.hip")))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DIMG_STAMPS", "-I",
           os.path.join(ROOT, "diffusion-model_amd/csrc"), "-I", os.path.join(ROOT, "include"), "-o", LIB] + src
    sys.exit(subprocess.call(cmd))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import dm_amd_bootstrap  # noqa: E402,F401
from dm_amd.runtime import hip, ops, pack  # noqa: E402

hip.LIB_PATH = LIB
hip._lib = None
lib = hip.lib()
lib.ddpm_img_set_dbg.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
N = 256
K33, K33S2, KT = hip.CONV_3X3_S1, hip.CONV_3X3_S2, hip.CONVT_2X2_S2
names = {1: "weights issued", 2: "loader done", 3: "barrier", 4: "mfma", 5: "tile->LDS", 6: "stats", 7: "stores", 8: "gnb: stats+halo+barrier",
         9: "gnb: pass1", 10: "gnb: combine1", 11: "gnb: S", 12: "gnb: pass2", 13: "gnb: combine2"}


def report(tag, run):
    run(); torch.cuda.synchronize()
    dbg = torch.zeros(4096 * 16, device=dev, dtype=torch.int64)
    lib.ddpm_img_set_dbg(dbg.data_ptr())
    run(); torch.cuda.synchronize()
    lib.ddpm_img_set_dbg(None)
    d = dbg.view(-1, 16).cpu()
    d = d[d[:, 0] > 0].double()
    order = [0, 1, 8, 9, 10, 11, 12, 13, 2, 3, 4, 5, 6, 7]
    order = [i for i in order if (d[:, i] > 0).all()]
    out = []
    for a_, b_ in zip(order[:-1], order[1:]):
        out.append(f"{names[b_]} {(d[:, b_] - d[:, a_]).mean() / 100:.2f}")
    life = (d[:, 7] - d[:, 0]).mean() / 100
    span = (d[:, 7].max() - d[:, 0].min()) / 100
    print(f"{tag}: workgroups {len(d)}, lifetime {life:.1f} us, kernel span {span:.1f} us\n    " + " | ".join(out), flush=True)


for tag, kind, ci, co, h in [("b 3x3 32->32 16^2", K33, 32, 32, 16), ("j convT 32->32 16^2", KT, 32, 32, 16), ("f convT 256->128 2^2", KT, 256, 128, 2),
                             ("a s2 32->32 32^2", K33S2, 32, 32, 32)]:
    k = 2 if kind == KT else 3
    x = torch.randn(N, h, h, ci, device=dev).bfloat16()
    w = torch.randn((ci, co, k, k) if kind == KT else (co, ci, k, k), device=dev) * 0.05
    wf, wd = pack.pack_fwd(kind, w), pack.pack_dgrad(kind, w)
    b = torch.randn(co, device=dev) * 0.1
    gamma, beta = torch.ones(co, device=dev), torch.zeros(co, device=dev)
    act = hip.ACT["gelu"]
    report(tag + " fwd", lambda: ops.conv_forward(kind, x, wf, b, co, gn=(gamma, beta, 16, act, None)))
    raw = ops.conv_forward(kind, x, wf, b, co)
    _, stats = ops.gn_act_forward(raw, gamma, beta, 16, act)
    dy = torch.randn_like(raw)
    part = torch.zeros(N, 3, co, device=dev)
    draw = torch.empty_like(raw)
    report(tag + " bwd", lambda: ops.conv_forward(pack.DGRAD_KIND[kind], dy, wd, None, ci, rows=ci, gnb=(raw, gamma, beta, 16, act, stats, part, draw)))
