import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dm_amd_bootstrap  # noqa
from bench import build_model
from dm_amd.losses.ELBO import ELBOLoss
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
m = build_model(dev, 1); m.train()
crit = ELBOLoss("mse")
opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True)
x0 = torch.rand(64, 1, 32, 32, device=dev) * 2 - 1
def step():
    opt.zero_grad(set_to_none=True)
    t = torch.randint(1, 121, (64,), device=dev)
    pred = m(m.q(x0, t), t)
    loss = crit(m, t, pred, x0)["Loss"]
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
