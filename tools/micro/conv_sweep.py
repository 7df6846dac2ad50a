"""Throughput of the matrix-core-bound convolution class (3x3 stride 1, Cin >= 128) at saturating batch sizes.

    python tools/micro/conv_sweep.py                      # the whole sweep, timed with HIP events, JSON on stdout
    python tools/micro/conv_sweep.py --one 128,128,16,2048,fwd,pipe --reps 20     # one case (what a counter pass profiles)

A case = (Cin, Cout, image width, batch, fwd | dgrad | wgrad, kernel): kernel `pipe` = the pipelined 256 x 128 tile kernel
(csrc/conv_pipe.hip), `old` = what the dispatcher picked before it existed (general tap-list / small-image kernels)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dm_amd_bootstrap  # noqa: E402,F401
from dm_amd.runtime import hip, ops, pack  # noqa: E402

dev = torch.device("cuda", 0)
K33 = hip.CONV_3X3_S1


def run_case(cin, cout, w, n, what, kernel, reps):
    torch.manual_seed(0)
    wt = (torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5)
    x = torch.randn(n, w, w, cin, device=dev).bfloat16()
    dy = torch.randn(n, w, w, cout, device=dev).bfloat16()
    bias = torch.randn(cout, device=dev) * 0.1
    wf, wd = pack.pack_fwd(K33, wt), pack.pack_dgrad(K33, wt)
    flops = 2.0 * n * w * w * cin * cout * 9
    prev = hip.lib().ddpm_conv_pipe_mode(1 if kernel == "pipe" else -1)
    try:
        if what == "fwd":
            out = torch.empty(n, w, w, cout, device=dev, dtype=torch.bfloat16)
            fn = lambda: ops.conv_forward(K33, x, wf, bias, cout, out=out)                      # noqa: E731
        elif what == "dgrad":
            out = torch.empty(n, w, w, cin, device=dev, dtype=torch.bfloat16)
            fn = lambda: ops.conv_forward(K33, dy, wd, None, cin, out=out)                      # noqa: E731
        else:
            gw = torch.zeros(9, cout, cin, device=dev)
            fn = lambda: ops.conv_wgrad(K33, x, dy, cout, gw)                                   # noqa: E731
        for _ in range(3):
            fn()
        used = hip.lib().ddpm_conv_last_kernel() if what != "wgrad" else -1
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
    finally:
        hip.lib().ddpm_conv_pipe_mode(prev)
    return {"cin": cin, "cout": cout, "image": f"{w}x{w}", "batch": n, "what": what, "kernel": kernel, "kernel_used": hip.CONV_KERNEL_NAMES.get(used, "conv_wgrad_pipe_kernel" if kernel == "pipe" else "conv_wgrad_kernel"),
            "us": us, "gflop": flops * 1e-9, "tflops": flops / us * 1e-6, "frac_of_2500_tflops": flops / us * 1e-6 / 2500.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--one", default="")
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    if a.one:
        cin, cout, w, n, what, kernel = a.one.split(",")
        print(json.dumps(run_case(int(cin), int(cout), int(w), int(n), what, kernel, a.reps)))
        return
    rows = []
    for cin, cout, w in ((128, 128, 16), (256, 256, 8), (256, 512, 4), (512, 256, 4)):
        for n in ((1024, 4096) if w == 16 else (4096, 16384) if w == 8 else (16384, 65536)):
            for what in ("fwd", "dgrad", "wgrad"):
                for kernel in ("pipe", "old"):
                    r = run_case(cin, cout, w, n, what, kernel, a.reps)
                    rows.append(r)
                    print(f"{cin:4d}->{cout:4d} {w:2d}x{w:<2d} N={n:6d} {what:5s} {kernel:4s} {r['kernel_used']:18s} {r['us']:9.1f} us {r['tflops']:8.1f} TFLOP/s", file=sys.stderr, flush=True)
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
