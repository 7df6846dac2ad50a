"""Time every launch shape of one eager training step on its own (the probe of runtime.roofline), largest total first.
python tools/micro/step_shapes.py [filter substring] [out file]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import dm_amd_bootstrap  # noqa
from dm_amd.losses.ELBO import ELBOLoss
from dm_amd.neural_nets.denoise_models import ddpm as dm_ddpm, unets as dm_unets
from dm_amd.runtime import roofline

YAML = dict(stages_channels=[32, 64, 128, 256], expand_factor=2, drop_p=0.1, T=120, beta=0.25, t_embed_dim=32, activation="gelu",
            base_groups_norm=4, initializer="he_normal", dtype=None)
dev = torch.device("cuda:0")
m = dm_ddpm.DDPM(dm_unets.BaseUNet, **YAML, is_conditional=False, img_channels=1, reconstruction_method="mse", latent_dim=512, device=dev).to(dev)
m.train()
crit = ELBOLoss("mse")
x0 = torch.rand(256, 1, 32, 32, device=dev) * 2 - 1


def step():
    t = torch.randint(1, 121, (256,), device=dev)
    pred = m(m.q(x0, t), t)
    crit(m, t, pred, x0)["Loss"].backward()
    m.zero_grad(set_to_none=True)


rows = []
roofline.measure(step, detail=rows)
flt = sys.argv[1] if len(sys.argv) > 1 else ""
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/step_shapes.txt"
with open(out, "w") as f:
    tot = 0.0
    for k, cnt, us, gbs, tf in sorted(rows, key=lambda r: -r[1] * r[2]):
        if flt in k:
            tot += cnt * us
            f.write(f"{k:72s} x{cnt:3d} {us:8.1f} us {gbs:8.1f} GB/s {tf:7.1f} TF/s  {cnt * us:8.1f} us/step\n")
    f.write(f"total {tot:.1f} us\n")
print(open(out).read())
