"""Host time spent inside PipelinedTrainStep.replay() against the device time of a step: tells whether the step is bound by
graph-launch work on the host (python tools/micro/host_time.py)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bench import ELBOLoss, FusedAdam, build_model, graphs  # noqa: E402

dev = torch.device("cuda", 0)
m = build_model(dev, 1)
m.train()
opt = FusedAdam(m.parameters(), lr=1e-3)
x0 = torch.rand(256, 1, 32, 32, device=dev) * 2 - 1
step = graphs.PipelinedTrainStep(m, ELBOLoss("mse"), opt, x0, None, phases=4)
for _ in range(20):
    step.replay()
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    step.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host time per replay {1e3 * (t1 - t0) / n:.3f} ms; device-complete per step {1e3 * (t2 - t0) / n:.3f} ms")
# replay with a sync after each step: no host run-ahead at all
t0 = time.perf_counter()
for _ in range(n):
    step.replay()
    torch.cuda.synchronize()
print(f"synchronised per step {1e3 * (time.perf_counter() - t0) / n:.3f} ms")
