"""One full sampling job (two concurrent 512-image chains, 32x32x1) and nothing else: run under rocprofv3 --kernel-trace --stats
to see where a sampling step's time goes (python tools/micro/sample_only.py [images per chain] [streams])."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import build_model, graphs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
m = build_model(dev, 1)
m.eval()
smp = graphs.ConcurrentSampler(m, n, (32, 32), streams) if streams > 1 else graphs.GraphedSampler(m, n, (32, 32))
smp.sample()
torch.cuda.synchronize()
t0 = time.perf_counter()
smp.sample()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{n * streams} images in {1e3 * dt:.1f} ms: {n * streams / dt:.0f} images/s, {1e3 * dt / 120:.3f} ms per step")
