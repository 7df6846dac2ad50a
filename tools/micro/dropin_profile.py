"""Host-side profile (cProfile) of the reference trainer's loop shape on the module surface: python tools/micro/dropin_profile.py"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bench import ELBOLoss, FusedAdam, build_model  # noqa: E402

dev = torch.device("cuda", 0)
m = build_model(dev, 1)
m.train()
crit = ELBOLoss("mse")
opt = FusedAdam(m.parameters(), lr=1e-3)
x0 = torch.rand(256, 1, 32, 32, device=dev) * 2 - 1


def micro(i):
    t = torch.randint(1, m.T + 1, (256,), device=dev)
    x_t = m.q(x0, t)
    loss = crit(m, t, m(x_t, t), x0, is_ddp=False)["Loss"]
    loss.backward(retain_graph=True)
    if i % 3 == 0:
        opt.step()
        opt.zero_grad(set_to_none=True)


for i in range(1, 10):
    micro(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(1, 31):
    micro(i)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
