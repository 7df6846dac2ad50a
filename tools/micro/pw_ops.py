"""The pointwise (1x1) convolutions and the normalise-on-load depthwise kernel of large images, timed alone (HIP events) with
their algorithmic GB/s, once with gelu in the loader and once without (tells arithmetic-bound from bandwidth-bound):
python tools/micro/pw_ops.py [batch]          (PWS_MODE=1 / -1: the streaming pointwise kernel everywhere / nowhere)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dm_amd_bootstrap  # noqa: E402,F401
from dm_amd.runtime import hip, ops, pack  # noqa: E402

dev = torch.device("cuda", 0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K11 = hip.CONV_1X1
if os.environ.get("PWS_MODE"):
    hip.lib().ddpm_conv_pw_stream_mode(int(os.environ["PWS_MODE"]))
G = 16


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


def line(tag, us, nbytes):
    print(f"{tag:58s} {us:8.1f} us {nbytes / us * 1e-3:7.0f} GB/s", flush=True)


for h, c, e in ((128, 32, 64), (64, 32, 64), (64, 64, 128), (32, 64, 128), (32, 128, 256)):
    n = nb
    x = torch.randn(n, h, h, c, device=dev).bfloat16()
    xe = torch.randn(n, h, h, e, device=dev).bfloat16()
    we = pack.pack_fwd(K11, torch.randn(e, c, 1, 1, device=dev) / c ** 0.5)
    wp = pack.pack_fwd(K11, torch.randn(c, e, 1, 1, device=dev) / e ** 0.5)
    be, bp = torch.randn(e, device=dev) * 0.1, torch.randn(c, device=dev) * 0.1
    w9 = torch.randn(9, e, device=dev)
    aff = torch.stack([torch.rand(n, e, device=dev) + 0.5, torch.randn(n, e, device=dev) * 0.1], dim=2).contiguous()
    oe, oc = torch.empty_like(xe), torch.empty_like(x)
    bx, bxe = x.numel() * 2, xe.numel() * 2
    tag = f"{h}x{h} {c}->{e} N{n}: "
    line(tag + "expansion", timed(lambda: ops.conv_forward(K11, x, we, be, e, out=oe)), bx + bxe)
    line(tag + "expansion + sums", timed(lambda: ops.conv_forward(K11, x, we, be, e, out=oe, gn_part_groups=G)), bx + bxe)
    line(tag + "projection", timed(lambda: ops.conv_forward(K11, xe, wp, bp, c, out=oc)), bx + bxe)
    for act in ("none", "gelu"):
        a = hip.ACT[act] if act != "none" else 0
        line(tag + f"projection, on-load {act} + sums",
             timed(lambda: ops.conv_forward(K11, xe, wp, bp, c, out=oc, gn_part_groups=G, in_affine=(aff, a))), bx + bxe)
        if ops.dwconv3x3_affine_supported(xe, G):
            line(tag + f"depthwise, on-load {act} + sums", timed(lambda: ops.dwconv3x3_affine(xe, aff, a, w9, be, G)), 2 * bxe)
    line(tag + "depthwise + sums", timed(lambda: ops.dwconv3x3(xe, w9, be, out=oe, gn_part_groups=G)), 2 * bxe)
    line(tag + "copy of the expanded tensor", timed(lambda: oe.copy_(xe)), 2 * bxe)
    del x, xe, oe, oc, aff
    torch.cuda.empty_cache()
