"""128x128x3 sampling with self-attention at the 16x16 stage, a few steps of the chain at several batch sizes:
python tools/micro/sample128.py [batch ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import build_model, graphs  # noqa: E402

dev = torch.device("cuda", 0)
m = build_model(dev, 3, [2])
m.eval()
for nb in [int(v) for v in sys.argv[1:]] or [128, 256, 512]:
    smp = graphs.GraphedSampler(m, nb, (128, 128), use_graph=os.environ.get("NOGRAPH") != "1")     # NOGRAPH=1: plain launches (counter passes)
    smp.begin()
    for _ in range(3):
        smp.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        smp.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"batch {nb}: {1e3 * dt:.2f} ms per step, {nb / (120 * dt):.1f} images/s over a 120-step chain", flush=True)
    del smp
    torch.cuda.empty_cache()
