"""Run-to-run reproducibility of the model's parameter gradients (same inputs, same weights, two backward passes)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import dm_amd_bootstrap  # noqa
from test_gpu_unet import build, ELBOLoss, hash_uniform, hash_normal
dev = torch.device("cuda:0")
m, _, _ = build(dev, 3, "gelu")
m.eval()
crit = ELBOLoss("mse")
x0 = (hash_uniform((4, 3, 32, 32), 5) * 2 - 1).to(dev)
t = torch.tensor([3, 40, 80, 120], device=dev)
noise = hash_normal((4, 3, 32, 32), 50).to(dev)
gs = []
for fuse in (True, True, False, False):
    m.model._get_engine().fuse_sir = fuse
    m.zero_grad(set_to_none=True)
    pred = m(m.q(x0, t, noise=noise), t)
    crit(m, t, pred, x0)["Loss"].backward()
    gs.append({k: p.grad.clone() for k, p in m.named_parameters()})
def cmp(a, b, tag):
    worst = sorted(((((a[k] - b[k]).norm() / (b[k].norm() + 1e-20)).item(), k) for k in a), reverse=True)[:5]
    print(tag, [(f"{v:.2e}", k[-45:]) for v, k in worst])
cmp(gs[0], gs[1], "fused vs fused  :")
cmp(gs[2], gs[3], "unfused vs unfused:")
cmp(gs[0], gs[2], "fused vs unfused:")
