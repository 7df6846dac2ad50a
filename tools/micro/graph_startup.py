"""Start-up cost of a HIP graph replayed behind another graph on the same stream, against its node count and against
splitting it in two (python tools/micro/graph_startup.py).  Every node is the same ~20 us elementwise kernel, so
time per replay - nodes * kernel time = what the graph boundary costs."""
import time

import torch

dev = torch.device("cuda", 0)
x = torch.zeros(1 << 24, device=dev)


def body(n):
    for _ in range(n):
        x.add_(1.0)


def capture(n):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(n)
    return g


def run(graphs, reps=200):
    for g in graphs:
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for g in graphs:
            g.replay()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    body(8)
    torch.cuda.synchronize()
    g1 = capture(1)
    k = run([g1] * 50, 20) / 50
    per_kernel = None
    for n in (10, 35, 70, 140):
        g = capture(n)
        t = run([g])
        if per_kernel is None:
            big = capture(400)
            per_kernel = (run([big], 20) - t) / (400 - n)
        print(f"one graph of {n:3d} nodes: {t:8.1f} us per replay, kernel {per_kernel:.1f} us -> boundary {t - n * per_kernel:6.1f} us")
    for parts in ((35, 35), (5, 65), (10, 30, 30)):
        gs = [capture(n) for n in parts]
        t = run(gs)
        print(f"graphs of {parts} nodes back to back: {t:8.1f} us per round -> boundaries {t - sum(parts) * per_kernel:6.1f} us")
