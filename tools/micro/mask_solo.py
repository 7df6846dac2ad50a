"""Solo duration of the dropout-bit kernel for 22 MB of keep bits (the benchmarked step's amount): python tools/micro/mask_solo.py"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
hip = importlib.import_module("diffusion-model_amd.runtime.hip")
dev = torch.device("cuda", 0)
total = 22 * 1024 * 1024
offs = torch.tensor([0, total // 2, total], device=dev, dtype=torch.long)
seed = torch.tensor([77], device=dev, dtype=torch.long)
m = torch.zeros(total, device=dev, dtype=torch.uint8)
run = lambda: hip.call("ddpm_dropout_masks", hip.ptr(m), hip.ptr(offs), 2, total, 0.9, hip.ptr(seed), hip.stream())  # noqa: E731
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    run()
e1.record()
torch.cuda.synchronize()
print(os.environ.get("DDPM_MASK_GRID"), f"{e0.elapsed_time(e1) * 1e3 / 50:.1f} us")
