"""Solo timings of single weight-gradient problems of the benchmarked network (batch 256), through ddpm_conv_wgrad with
partial-sum rows: python tools/micro/wgrad_bench.py [path of another build of the library]."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib  # noqa: E402

pkg = importlib.import_module("diffusion-model_amd.runtime.hip")
if len(sys.argv) > 1:
    pkg.LIB_PATH = os.path.abspath(sys.argv[1])
ops = importlib.import_module("diffusion-model_amd.runtime.ops")
hip = pkg
dev = torch.device("cuda", 0)
N = 256
# kind, H, C0, C1, Cout, in_add, mask, tiles per split
CASES = [
    (0, 32, 32, 0, 32, False, False, 8), (0, 32, 32, 0, 32, False, False, 2), (0, 32, 16, 0, 32, False, False, 2),
    (0, 32, 32, 0, 1, False, False, 8),
    (0, 16, 32, 32, 32, True, False, 8), (0, 8, 64, 64, 64, True, False, 8), (0, 4, 128, 128, 128, True, False, 8),
    (0, 2, 256, 256, 256, True, False, 8), (0, 2, 512, 0, 256, False, False, 8),
    (1, 32, 32, 0, 32, True, False, 2), (1, 16, 32, 0, 64, True, False, 8), (1, 8, 64, 0, 128, True, False, 8),
    (3, 16, 32, 0, 32, False, False, 8), (3, 8, 64, 0, 32, False, False, 8), (3, 4, 128, 0, 64, False, False, 8),
    (2, 32, 64, 0, 32, False, False, 8), (2, 32, 32, 0, 64, False, True, 8), (2, 16, 64, 0, 32, False, False, 8),
    (2, 16, 64, 0, 32, False, False, 2), (2, 8, 128, 0, 64, False, False, 8), (2, 4, 256, 0, 128, False, False, 8),
    (2, 2, 512, 0, 256, False, False, 8), (2, 2, 256, 0, 512, False, True, 8),
]
KINDS = [int(k) for k in os.environ.get("WG_KINDS", "0,1,2,3").split(",")]
for kind, h, c0, c1, cout, add, mask, tps in CASES:
    if kind not in KINDS:
        continue
    oh = h // 2 if kind == 1 else h * 2 if kind == 3 else h
    x0 = torch.randn(N, h, h, c0, device=dev).bfloat16()
    x1 = torch.randn(N, h, h, c1, device=dev).bfloat16() if c1 else None
    cs = (cout + 7) // 8 * 8
    dy = torch.randn(N, oh, oh, cs, device=dev).bfloat16()
    in_add = torch.randn(N, c0, device=dev) if add else None
    in_mask = torch.randint(0, 256, (N * h * h * c0 // 8,), device=dev, dtype=torch.uint8) if mask else None
    splits = ops.conv_wgrad_splits(kind, x0, dy, cout, x1=x1, tiles_per_split=tps)
    slices = {0: 9, 1: 9, 2: 1, 3: 4}[kind]
    part = torch.zeros(max(splits, 1) * slices * cout * (c0 + c1), device=dev)
    dw = torch.zeros(slices, cout, c0 + c1, device=dev)
    kw = dict(x1=x1, in_add=in_add, in_mask=in_mask, in_scale=1.1, tiles_per_split=tps)
    run = (lambda: ops.conv_wgrad(kind, x0, dy, cout, None, part=part, **kw)) if splits > 1 else (lambda: ops.conv_wgrad(kind, x0, dy, cout, dw, **kw))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    # 20 launches replayed as one graph: the host's per-launch cost (~10 us through ctypes) stays out of the measurement
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    mb = 2 * N * (h * h * (c0 + c1) + oh * oh * cs) / 1e6
    print(f"kind {kind} {h:2d}x{h:<2d} Cin {c0:3d}+{c1:<3d} Cout {cout:3d} add {int(add)} mask {int(mask)} tps {tps} splits {splits:4d}: "
          f"{us:7.1f} us  {mb:6.1f} MB  {mb / us / 1e3 * 1e3:7.1f} GB/s", flush=True)
