"""Lists the weight-gradient problems of every backward phase of the benchmarked training step: shapes, position splits,
workgroups and the bytes each problem has to read (python tools/micro/wgrad_list.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402,F401
from bench import ELBOLoss, FusedAdam, build_model, graphs  # noqa: E402

eng_mod = sys.modules[graphs.__name__.rsplit(".", 1)[0] + ".engine"]
hip = eng_mod.hip
orig = eng_mod.UNetEngine._launch_wgrad_group
launch = [0]


def listed(self, pending, dev):
    launch[0] += 1
    print(f"--- group launch {launch[0]}: {len(pending)} problems")
    tot = 0
    for p in pending:
        d = p[0]
        splits = int(hip.lib().ddpm_conv_wgrad_splits(__import__("ctypes").byref(d)))
        cin = d.C0 + d.C1
        oh = d.IH // 2 if d.kind == 1 else d.IH * 2 if d.kind == 3 else d.IH
        rd = 2 * d.N * (d.IH * d.IW * cin + oh * oh * d.dy_cs)
        tot += rd
        print(f"kind {d.kind} N {d.N} {d.IH}x{d.IW} Cin {d.C0}+{d.C1} Cout {d.Cout} (dy stride {d.dy_cs}) tps {d.tiles_per_split} "
              f"splits {splits} mask {bool(d.in_mask)} add {bool(d.in_add)} part {bool(d.part)} reads {rd / 1e6:.1f} MB")
    print(f"    total reads {tot / 1e6:.1f} MB")
    return orig(self, pending, dev)


eng_mod.UNetEngine._launch_wgrad_group = listed
dev = torch.device("cuda", 0)
m = build_model(dev, 1)
m.train()
opt = FusedAdam(m.parameters(), lr=1e-3)
x0 = torch.rand(256, 1, 32, 32, device=dev) * 2 - 1
step = graphs.PipelinedTrainStep(m, ELBOLoss("mse"), opt, x0, None, phases=4, use_graph=False, warmup=1)
