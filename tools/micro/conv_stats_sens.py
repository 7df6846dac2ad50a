"""how far a one-bf16-step perturbation of the input moves the 128x128 network's prediction (conv-side sums off)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["DDPM_MODULE_GRAPHS"] = "0"
from bench import build_model  # noqa: E402

dev = torch.device("cuda", 0)
m = build_model(dev, 3, [2])
m.eval()
eng = [mod for mod in m.modules() if hasattr(mod, "_get_engine")][0]._get_engine()
n = 4
x = torch.randn(n, 3, 128, 128, device=dev)
t = torch.full((n,), m.T, device=dev, dtype=torch.long)


def l2(a, b):
    return ((a - b).norm() / b.norm()).item()


with torch.no_grad():
    eng.conv_stats = False
    base = m(x, t).float().clone()
    for eps in (1e-4, 1e-3, 4e-3):
        print("input noise", eps, "->", l2(m(x + eps * torch.randn_like(x), t).float(), base), flush=True)
    eng.conv_stats = True
    print("conv-side sums ->", l2(m(x, t).float(), base), flush=True)
    for tt in (1, 30, 60):
        t2 = torch.full((n,), tt, device=dev, dtype=torch.long)
        eng.conv_stats = False
        b2 = m(x, t2).float().clone()
        eng.conv_stats = True
        print("t", tt, "conv-side sums ->", l2(m(x, t2).float(), b2), flush=True)
