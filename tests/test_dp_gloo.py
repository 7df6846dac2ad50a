"""Data-parallel gradient exchange on two CPU ranks (gloo): the flat-view fast path, the gather/scatter path, parameter
broadcast and the sampling shard split.  Mirrors what the reference gets from torch's DistributedDataParallel wrapper
(base_trainer.py:95-100 there): after the exchange every rank holds the mean gradient."""
import os
import socket

import numpy as np
import torch
import torch.distributed as td
import torch.multiprocessing as mp

import dm_amd_bootstrap  # noqa: F401
from dm_amd.runtime.dp import GradReducer, sample_sharded, shard_range

SHAPES = [(5, 3), (7,), (2, 3, 3, 3), (1,)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)
        params = [torch.nn.Parameter(torch.randn(s)) for s in SHAPES]
        red = GradReducer(params, world, bucket_bytes=64)            # 16-element buckets: several collectives
        red.broadcast_parameters()
        sums = [p.detach().sum().item() for p in params]

        # (1) gradients as views of one flat buffer, the layout the HIP engine's backward produces
        flat = torch.zeros(red.total)
        for p, o in zip(params, red.offs):
            p.grad = flat[o:o + p.numel()].view(p.shape)
            p.grad.fill_(float(rank + 1))
        fv = red.flat_view()
        assert fv is not None and fv.data_ptr() == flat.data_ptr() and len(red.buckets(fv)) > 1
        red.allreduce()
        ok1 = all(torch.allclose(p.grad, torch.full_like(p, (1 + world) / 2)) for p in params)

        # (2) independent gradient tensors, one of them missing
        for i, p in enumerate(params):
            p.grad = None if i == 1 else torch.full_like(p, float(10 * (rank + 1) + i))
        assert red.flat_view() is None
        red.allreduce()
        ok2 = all(torch.allclose(p.grad, torch.full_like(p, 0.0 if i == 1 else 10 * (1 + world) / 2 + i)) for i, p in enumerate(params))
        # (3) the overlapped path: two halves of the flat buffer exchanged as separate asynchronous ranges
        flat.copy_(torch.arange(red.total, dtype=torch.float32) * (rank + 1))
        cut = red.offs[2]
        h_hi = red.start_range(flat, cut, red.total)
        h_lo = red.start_range(flat, 0, cut)
        red.finish(h_hi)
        red.finish(h_lo)
        ok3 = torch.allclose(flat, torch.arange(red.total, dtype=torch.float32) * (1 + world) / 2)
        # (4) sharded sampling: every rank produces its share (here: images filled with their global index), rank 0 receives all
        total = 5
        b0, _ = shard_range(total, rank, world)
        firsts = []

        def fake(n, first):
            firsts.append(first)
            return (torch.arange(n, dtype=torch.float32) + first).view(n, 1, 1, 1).expand(n, 1, 2, 2).contiguous()

        imgs = sample_sharded(fake, total)
        ok4 = firsts == [b0] and ((imgs is None) if rank else (imgs.shape == (total, 1, 2, 2) and
                                                                torch.equal(imgs[:, 0, 0, 0], torch.arange(total, dtype=torch.float32))))
        # (5) noise keyed by GLOBAL image index makes the job independent of the world size: a stand-in chain (x_T and three
        # noisy steps, all Philox draws at the global element index, the keying the HIP sampler uses) sharded over the ranks
        # equals the same chain computed in one piece
        from oracle import philox
        chw = 8

        def chain(n, first, seed=77):
            x = philox.normal(n * chw, seed, 1, first * chw)
            for k in range(3):
                x = np.float32(0.9) * x + np.float32(0.1) * philox.normal(n * chw, seed + 1000 * (k + 1), 2, first * chw)
            return torch.from_numpy(x).view(n, 2, 2, 2)

        got = sample_sharded(chain, total)
        ok4 = ok4 and ((got is None) if rank else torch.equal(got, chain(total, 0)))
        q.put((rank, sums, ok1, ok2 and ok3 and ok4))
    finally:
        td.destroy_process_group()


def test_two_rank_gradient_exchange():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1], "broadcast did not equalise the parameters"
    assert all(r[2] and r[3] for r in res)


def _grad_of_rank(n, rank):
    g = torch.Generator().manual_seed(500 + rank)
    return torch.randn(n, generator=g) * torch.logspace(-6, 2, n)        # eight decades of magnitudes, like a network's gradients


def _wire_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params = [torch.nn.Parameter(torch.zeros(s)) for s in SHAPES]
        red = GradReducer(params, world, bucket_bytes=32, wire="bf16")     # 16 bf16 elements per collective
        flat = torch.zeros(red.total)
        for p, o in zip(params, red.offs):
            p.grad = flat[o:o + p.numel()].view(p.shape)
        flat.copy_(_grad_of_rank(red.total, rank))
        assert len(red.buckets(flat.bfloat16())) > 1
        # what every rank must end with: both ranks' gradients rounded to bf16, summed and rounded to bf16 once more (the
        # collective's arithmetic type), widened to fp32 exactly, then halved in fp32
        summed = (_grad_of_rank(red.total, 0).bfloat16() + _grad_of_rank(red.total, 1).bfloat16()).float() * 0.5
        cut = red.offs[2]
        h_hi = red.start_range(flat, cut, red.total)
        h_lo = red.start_range(flat, 0, cut)
        red.finish(h_hi)
        red.finish(h_lo)
        ok_range = torch.equal(flat, summed)
        exact = torch.equal((flat * 2).bfloat16().float(), flat * 2)      # every value is a widened bf16 (times the exact 1/2)
        flat.copy_(_grad_of_rank(red.total, rank))
        red.allreduce()                                                    # the one-call path takes the same wire
        ok_all = torch.equal(flat, summed) and all(p.grad.data_ptr() == flat[o:].data_ptr() for p, o in zip(params, red.offs))
        fp32 = (_grad_of_rank(red.total, 0) + _grad_of_rank(red.total, 1)) * 0.5
        rel = ((flat - fp32).abs() / fp32.abs().clamp_min(1e-30)).max().item()
        q.put((rank, ok_range, exact, ok_all, rel, flat.clone()))
    finally:
        td.destroy_process_group()


def test_two_rank_bf16_wire_format_reexpands_to_the_same_fp32_on_every_rank():
    """The optional bf16 wire halves the bytes on the links; the fp32 gradients every rank ends with are bit-equal across
    ranks, equal to the bf16-rounded sum widened exactly, and within three bf16 roundings (3 * 2^-9) of the fp32 exchange
    wherever the two ranks' terms do not cancel."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_wire_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] and r[3] for r in res), [(r[1], r[2], r[3]) for r in res]
    assert torch.equal(res[0][5], res[1][5])
    a, b = _grad_of_rank(res[0][5].numel(), 0), _grad_of_rank(res[0][5].numel(), 1)
    fp32 = (a + b) * 0.5
    bound = 3 * 2.0 ** -9 * (a.abs() + b.abs()) * 0.5
    assert bool(((res[0][5] - fp32).abs() <= bound).all())


def test_shard_range_covers_everything_once():
    for total in (0, 1, 7, 256, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
