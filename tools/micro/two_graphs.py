"""Do two captured graphs replayed on two streams run concurrently?  Each graph = a chain of L small dependent kernels."""
import time, torch
dev = torch.device("cuda")
L = 200
def chain_graph(x, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        for _ in range(L):
            x.add_(1.0)
    return g
for n in (1 << 12, 1 << 20, 1 << 22):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    x1, x2 = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    g1, g2 = chain_graph(x1, s1), chain_graph(x2, s2)
    def both(concurrent):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            if concurrent:
                with torch.cuda.stream(s1): g1.replay()
                with torch.cuda.stream(s2): g2.replay()
            else:
                with torch.cuda.stream(s1): g1.replay(); g2.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 10 * 1e6
    both(True); both(False)
    print(f"n={n:8d}: sequential {both(False):8.1f} us, two streams {both(True):8.1f} us", flush=True)
    # eager kernels on two streams for comparison
    def eager(concurrent):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s, x in ((s1, x1), (s2 if concurrent else s1, x2)):
            with torch.cuda.stream(s):
                for _ in range(L):
                    x.add_(1.0)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e6
    eager(True); eager(False)
    print(f"            eager: sequential {eager(False):8.1f} us, two streams {eager(True):8.1f} us", flush=True)
