"""How many hardware queues execute kernels at the same time?  K streams, each bound to a queue by a first launch, each running a
chain of small dependent kernels; read the overlap from a rocprofv3 --kernel-trace with tools/micro/queue_overlap.py.
usage: python tools/micro/multi_queue.py [K] [graph]"""
import sys
import time

import torch

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
use_graph = len(sys.argv) > 2 and sys.argv[2] == "graph"
dev = torch.device("cuda:0")
streams, xs = [], []
for _ in range(K):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        xs.append(torch.zeros(1 << 16, device=dev))
    torch.cuda.synchronize()
    streams.append(st)


def chain(x):
    for _ in range(50):
        x.mul_(1.0001).add_(1e-3)


graphs = []
if use_graph:
    for st, x in zip(streams, xs):
        with torch.cuda.stream(st):
            chain(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            chain(x)
        graphs.append(g)
torch.cuda.synchronize()
for k in (1, K):
    t0 = time.perf_counter()
    for _ in range(20):
        for i in range(k):
            with torch.cuda.stream(streams[i]):
                graphs[i].replay() if use_graph else chain(xs[i])
    torch.cuda.synchronize()
    print(f"{k} stream(s): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per round of {k} chains of 100 kernels", flush=True)
