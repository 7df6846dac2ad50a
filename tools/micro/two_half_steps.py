"""Experiment: does the machine take two half-batch training steps side by side faster than one full-batch step?
Two independent models, each with its own pipelined step at batch B/2 on its own pair of streams, against one at batch B.
usage: python tools/micro/two_half_steps.py [B] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from bench import ELBOLoss, FusedAdam, build_model, graphs  # noqa: E402


def make(dev, n, main, side):
    m = build_model(dev, 1)
    m.train()
    opt = FusedAdam(m.parameters(), lr=1e-3)
    x0 = torch.rand(n, 1, 32, 32, device=dev) * 2 - 1
    with torch.cuda.stream(main):
        st = graphs.PipelinedTrainStep(m, ELBOLoss("mse"), opt, x0, None, side_stream=side)
    torch.cuda.synchronize()
    return st


def timeit(fn, steps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    dev = torch.device("cuda:0")
    pool = [graphs._pool_stream(dev, k) for k in range(4)]
    if os.environ.get('BIND', '0') == '1':      # fresh streams, each bound to a hardware queue by a first launch, in creation order
        pool = []
        for _ in range(4):
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                torch.zeros(64, device=dev)
            torch.cuda.synchronize()
            pool.append(st)
    if os.environ.get('HIPRIO', '1') == '1':
        pool[2], pool[3] = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=-1)
    one = make(dev, B, pool[1], pool[0])

    def f1():
        with torch.cuda.stream(pool[1]):
            one.replay()
    t1 = timeit(f1, steps)
    print(f"one step of batch {B}: {t1 * 1e3:.3f} ms  {B / t1:.0f} img/s", flush=True)
    ha = make(dev, B // 2, pool[1], pool[0])
    t_half = timeit(lambda: [torch.cuda.stream(pool[1]).__enter__(), ha.replay(), torch.cuda.stream(pool[1]).__exit__(None, None, None)], steps) if False else None
    def fh():
        with torch.cuda.stream(pool[1]):
            ha.replay()
    th = timeit(fh, steps)
    print(f"one step of batch {B // 2} alone: {th * 1e3:.3f} ms  {B / 2 / th:.0f} img/s", flush=True)
    hb = make(dev, B // 2, pool[2], pool[3])

    def f2():
        with torch.cuda.stream(pool[1]):
            ha.replay()
        with torch.cuda.stream(pool[2]):
            hb.replay()
    t2 = timeit(f2, steps)
    torch.cuda.synchronize()
    ta = tb = 0.0
    for _ in range(steps):
        h0 = time.perf_counter()
        with torch.cuda.stream(pool[1]):
            ha.replay()
        h1 = time.perf_counter()
        with torch.cuda.stream(pool[2]):
            hb.replay()
        h2 = time.perf_counter()
        ta += h1 - h0; tb += h2 - h1
    torch.cuda.synchronize()
    print(f"host time per replay: A {1e3 * ta / steps:.3f} ms, B {1e3 * tb / steps:.3f} ms", flush=True)
    print(f"two steps of batch {B // 2} side by side: {t2 * 1e3:.3f} ms  {B / t2:.0f} img/s", flush=True)


if __name__ == "__main__":
    main()
