"""Does a captured graph run parallel branches concurrently?  K chains of L small dependent kernels each."""
import sys, time, torch
dev = torch.device("cuda")
L = 200
def run(K, n):
    xs = [torch.zeros(n, device=dev) for _ in range(K)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        for s, x in zip(streams, xs):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                for _ in range(L):
                    x.add_(1.0)
        for s in streams:
            main.wait_stream(s)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e6
for n in (1 << 12, 1 << 20, 1 << 22):
    for K in (1, 2, 4, 8):
        us = run(K, n)
        print(f"n={n:8d} K={K}: {us:8.1f} us total, {us / L:6.2f} us per chain link, {us / (L * K):6.2f} us per kernel", flush=True)
