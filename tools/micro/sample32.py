"""32x32x1 sampling (the headline configuration's sampler), a few steps of the chain at several batch sizes:
python tools/micro/sample32.py [batch ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import build_model, graphs  # noqa: E402

dev = torch.device("cuda", 0)
m = build_model(dev, 1)
m.eval()
for nb in [int(v) for v in sys.argv[1:]] or [512, 1024]:
    smp = graphs.GraphedSampler(m, nb, (32, 32))
    smp.begin()
    for _ in range(5):
        smp.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        smp.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    print(f"batch {nb}: {1e3 * dt:.3f} ms per step, {nb / (120 * dt):.1f} images/s over a 120-step chain", flush=True)
    del smp
    torch.cuda.empty_cache()
