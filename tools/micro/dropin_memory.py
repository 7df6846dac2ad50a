"""Memory of a long run of the reference trainer's loop on the module surface (the module's captured graphs must not grow): python tools/micro/dropin_memory.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from bench import ELBOLoss, FusedAdam, build_model
dev = torch.device("cuda", 0)
m = build_model(dev, 1); m.train()
crit = ELBOLoss("mse"); opt = FusedAdam(m.parameters(), lr=1e-3)
x0 = torch.rand(256, 1, 32, 32, device=dev) * 2 - 1
for i in range(1, 301):
    t = torch.randint(1, m.T + 1, (256,), device=dev)
    loss = crit(m, t, m(m.q(x0, t), t), x0, is_ddp=False)["Loss"]
    loss.backward(retain_graph=True)
    if i % 3 == 0:
        opt.step(); opt.zero_grad(set_to_none=True)
    if i in (12, 100, 200, 300):
        torch.cuda.synchronize()
        print(i, round(torch.cuda.memory_allocated() / 2**20), "MiB allocated", round(torch.cuda.memory_reserved() / 2**20), "MiB reserved", "loss", round(loss.item(), 5))
