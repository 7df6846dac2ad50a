"""The image stem (3 of 16 stored channels -> 32, second activated result) and head (32 -> 3, tanh, fp32 NCHW) timed alone:
python tools/micro/thin_time.py [batch]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dm_amd_bootstrap  # noqa
from dm_amd.runtime import hip, ops, pack
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 100


for h in (128, 64, 32):
    x = torch.randn(n, h, h, 32, device=dev).bfloat16()
    w = pack.pack_fwd(hip.CONV_3X3_S1, torch.randn(3, 32, 3, 3, device=dev) * 0.05)
    b = torch.zeros(3, device=dev)
    t_head = timed(lambda: ops.conv_forward(hip.CONV_3X3_S1, x, w, b, 3, act=hip.ACT["tanh"], f32_nchw=True))
    xi = torch.zeros(n, h, h, 16, device=dev, dtype=torch.bfloat16)
    xi[..., :3] = torch.randn(n, h, h, 3, device=dev).bfloat16()
    wi = torch.zeros(32, 16, 3, 3, device=dev)
    wi[:, :3] = torch.randn(32, 3, 3, 3, device=dev) * 0.2
    wi = pack.pack_fwd(hip.CONV_3X3_S1, wi)
    bi = torch.zeros(32, device=dev)
    t_stem = timed(lambda: ops.conv_forward(hip.CONV_3X3_S1, xi, wi, bi, 32, act2=hip.ACT["gelu"], c0_used=3))
    t_stem_inf = timed(lambda: ops.conv_forward(hip.CONV_3X3_S1, xi, wi, bi, 32, act=hip.ACT["gelu"], c0_used=3))
    print(f"{h}x{h} N{n}: head {t_head:.1f} us, stem (training form, two results) {t_stem:.1f} us, stem (inference) {t_stem_inf:.1f} us", flush=True)
