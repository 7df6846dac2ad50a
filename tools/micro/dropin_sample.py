"""The reference's own sampling call on the drop-in module (DDPM.sample_tensor: the eager T-step loop through the module
surface), against the graph sampler: python tools/micro/dropin_sample.py [images]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import build_model  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
m = build_model(dev, 1)
m.eval()
for graphs_on in (False, True):
    m.model.use_graphs = graphs_on
    m.sample_tensor(n, (32, 32), steps=6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.sample_tensor(n, (32, 32))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"module graphs {graphs_on}: {n} images in {1e3 * dt:.1f} ms: {n / dt:.0f} images/s, {1e3 * dt / m.T:.3f} ms per step")
