"""How long the host takes to enqueue one pipelined training step (graph launches and events), against the GPU's step time.
usage: python tools/micro/host_enqueue.py [B ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import ELBOLoss, FusedAdam, build_model, graphs  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for B in [int(v) for v in sys.argv[1:]] or [256, 128, 16]:
        m = build_model(dev, 1)
        m.train()
        opt = FusedAdam(m.parameters(), lr=1e-3)
        x0 = torch.rand(B, 1, 32, 32, device=dev) * 2 - 1
        st = graphs.PipelinedTrainStep(m, ELBOLoss("mse"), opt, x0, None)
        for _ in range(5):
            st.replay()
        torch.cuda.synchronize()
        steps = 40
        t0 = time.perf_counter()
        for _ in range(steps):
            st.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"batch {B}: host enqueue {1e3 * (t1 - t0) / steps:.3f} ms/step, wall {1e3 * (t2 - t0) / steps:.3f} ms/step", flush=True)
        del st, m, opt


if __name__ == "__main__":
    main()
