"""Host-side cost of every call of one pipelined training step (graph replays, event records/waits): shows whether the
host keeps ahead of the GPU and which call blocks.  python tools/micro/host_times.py"""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import dm_amd_bootstrap  # noqa
from dm_amd.losses.ELBO import ELBOLoss
from dm_amd.neural_nets.denoise_models import ddpm as dm_ddpm, unets as dm_unets
from dm_amd.optimizer.optimizer import FusedAdam
from dm_amd.runtime import graphs

YAML = dict(stages_channels=[32, 64, 128, 256], expand_factor=2, drop_p=0.1, T=120, beta=0.25, t_embed_dim=32, activation="gelu",
            base_groups_norm=4, initializer="he_normal", dtype=None)
dev = torch.device("cuda:0")
m = dm_ddpm.DDPM(dm_unets.BaseUNet, **YAML, is_conditional=False, img_channels=1, reconstruction_method="mse", latent_dim=512, device=dev).to(dev)
m.train()
opt = FusedAdam(m.parameters(), lr=1e-3)
x0 = torch.rand(256, 1, 32, 32, device=dev) * 2 - 1
step = graphs.PipelinedTrainStep(m, ELBOLoss("mse"), opt, x0, None, phases=4)
acc = collections.defaultdict(float)


def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        acc[name] += time.perf_counter() - t
        return r
    return w


gm, gw, gc, gp = step.graphs
gc = gc or type("N", (), {"replay": lambda self: None})()
for i, g in enumerate(gm):
    g.replay = timed(f"M{i}.replay", g.replay)
for i, g in enumerate(gw):
    g.replay = timed(f"W{i}.replay", g.replay)
gc.replay = timed("C.replay", gc.replay)
if False:
    gp.replay = timed("P.replay", gp.replay)
for _ in range(5):
    step.replay()
torch.cuda.synchronize()
acc.clear()
N = 50
t0 = time.perf_counter()
for _ in range(N):
    step.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / N:.3f} ms/step; until GPU done {1e3 * (t2 - t0) / N:.3f} ms/step")
for k, v in sorted(acc.items()):
    print(f"   {k:12s} {1e6 * v / N:8.1f} us/step")
