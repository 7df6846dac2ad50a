"""Data-parallel gradient exchange: one process per GPU, flat all-reduces over RCCL (xGMI), overlapped with backward.

The reference wraps the model in stock torch DistributedDataParallel (base_trainer.py:95-100 there; one process per GPU
spawned by train_diff.py:114-126, NCCL group set up in experiments_setup/utils.py:170-187): per-parameter autograd hooks
feed 25 MB buckets.  Here the whole denoiser backward is one autograd node whose
350 parameter gradients come out as views of ONE flat fp32 buffer (27 MB for the 6.86 M-parameter network), so the
exchange is a single collective on that buffer with no packing pass; averaging is folded into the collective (AVG)
where the backend has it.  xGMI rings are per-link bound, so few large messages beat many small ones: `bucket_bytes`
defaults to one bucket for this network and only splits buffers larger than that.

Overlap: the engine's backward pass runs in phases (head-first); runtime/graphs.py's OverlappedTrainStep hands a range of
the flat buffer to start_range() as soon as the phase that completes it is done - the decoder half after the second of four
phases, the encoder's deep stages and latent layers after the third - so that those exchanges travel while the later phases
compute; only the exchange of the last phase's few parameters is exposed.

Gradients that are not views of one buffer (a foreign autograd graph, a CPU model under gloo) are gathered into the
reducer's own flat buffer, reduced there and scattered back."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as td
from torch import Tensor


def _pad4(n: int) -> int:
    return (n + 3) // 4 * 4


class GradReducer:
    def __init__(self, params: Sequence[Tensor], world: Optional[int] = None, *, bucket_bytes: int = 64 << 20, group=None,
                 wire: str = "fp32"):
        """wire = "bf16": the gradients travel as bf16 (half the bytes on the links: 13.7 MB instead of 27.4 MB for the
        denoiser) - each rank rounds its fp32 gradients to bf16 into a wire buffer, the collective sums / averages that
        buffer, and the result is widened back into the fp32 gradient buffer, so every rank ends with the same fp32 values,
        each exactly representable in bf16.  The sum itself is rounded to bf16 by the collective: one more rounding of
        2^-9 relative per gradient than the fp32 wire, which Adam's normalisation by the running second moment absorbs; the
        default stays fp32, which matches the reference's DistributedDataParallel bit for bit in what is summed."""
        assert wire in ("fp32", "bf16"), wire
        self.wire = wire
        self._wire_buf: Optional[Tensor] = None
        self.params: List[Tensor] = [p for p in params if p.requires_grad]
        self.group = group
        self.world = world if world is not None else td.get_world_size(group)
        self.bucket_bytes = int(bucket_bytes)
        self.offs, total = [], 0
        for p in self.params:
            self.offs.append(total)
            total += _pad4(p.numel())
        self.total = total
        self._own: Optional[Tensor] = None
        backend = td.get_backend(group) if td.is_initialized() else "none"
        self._avg = backend == "nccl"          # RCCL has ReduceOp.AVG; gloo does not

    # ------------------------------------------------------------------------------------------------------
    def flat_view(self) -> Optional[Tensor]:
        """The flat buffer all gradients are views of (laid out in parameter order, each padded to 4 elements), or None."""
        from .params import flat_grad_view
        return flat_grad_view(self.params, self.offs, self.total)

    def buckets(self, flat: Tensor) -> List[Tensor]:
        n, step = flat.numel(), max(1, self.bucket_bytes // flat.element_size())
        return [flat[i:min(n, i + step)] for i in range(0, n, step)]

    def _reduce(self, flat: Tensor) -> None:
        if self.wire == "bf16":
            self.finish(self.start_range(flat, 0, flat.numel()))
            return
        op = td.ReduceOp.AVG if self._avg else td.ReduceOp.SUM
        works = [td.all_reduce(b, op=op, group=self.group, async_op=True) for b in self.buckets(flat)]
        for w in works:
            w.wait()
        if not self._avg:
            flat.mul_(1.0 / self.world)

    @torch.no_grad()
    def start_range(self, flat: Tensor, lo: int, hi: int):
        """Begin averaging flat[lo:hi] over the ranks without blocking: the collective is ordered after the work already
        queued on the current stream and runs on the communicator's own stream, so kernels launched afterwards (the rest of
        the backward pass) overlap with it.  The range goes out in `bucket_bytes` pieces.  Returns a handle for finish()."""
        if self.world == 1:
            return None
        op = td.ReduceOp.AVG if self._avg else td.ReduceOp.SUM
        piece = flat[lo:hi]
        if self.wire == "bf16":
            if self._wire_buf is None or self._wire_buf.numel() != flat.numel() or self._wire_buf.device != flat.device:
                self._wire_buf = torch.empty(flat.numel(), device=flat.device, dtype=torch.bfloat16)
            w = self._wire_buf[lo:hi]
            w.copy_(piece)                        # round to nearest even, the same rounding the kernels use for activations
            return (piece, [td.all_reduce(b, op=op, group=self.group, async_op=True) for b in self.buckets(w)], w)
        return (piece, [td.all_reduce(b, op=op, group=self.group, async_op=True) for b in self.buckets(piece)], None)

    @torch.no_grad()
    def finish(self, handle) -> None:
        """Make the current stream wait for a start_range() exchange (host does not block on the GPU backend)."""
        if handle is None:
            return
        piece, works, wire = handle
        for w in works:
            w.wait()
        if wire is not None:
            piece.copy_(wire)                     # bf16 -> fp32 is exact
        if not self._avg:
            piece.mul_(1.0 / self.world)

    @torch.no_grad()
    def allreduce(self) -> None:
        """Average the gradients of all ranks in place."""
        if self.world == 1:
            return
        flat = self.flat_view()
        if flat is not None:
            self._reduce(flat)
            return
        p0 = self.params[0]
        if self._own is None:
            self._own = torch.zeros(self.total, device=p0.device, dtype=torch.float32)
        views = [self._own[o:o + p.numel()].view(p.shape) for p, o in zip(self.params, self.offs)]
        have = [(v, p.grad) for v, p in zip(views, self.params) if p.grad is not None]
        for v, p in zip(views, self.params):
            if p.grad is None:
                v.zero_()
        torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        self._reduce(self._own)
        for v, p in zip(views, self.params):
            if p.grad is None:
                p.grad = v.clone()
        torch._foreach_copy_([g for _, g in have], [v for v, _ in have])

    @torch.no_grad()
    def broadcast_parameters(self, src: int = 0) -> None:
        """Make every rank start from rank `src`'s weights (one flat broadcast)."""
        if self.world == 1:
            return
        ps = self.params
        flat = torch.cat([p.detach().reshape(-1).float() for p in ps])
        td.broadcast(flat, src, group=self.group)
        o = 0
        for p in ps:
            p.detach().copy_(flat[o:o + p.numel()].view(p.shape))
            o += p.numel()


def shard_range(total: int, rank: int, world: int):
    """Contiguous [begin, end) share of `total` independent units (images of a sampling job) for `rank`."""
    base, extra = divmod(total, world)
    b = rank * base + min(rank, extra)
    return b, b + base + (1 if rank < extra else 0)


@torch.no_grad()
def sample_sharded(sample_fn, total: int, group=None) -> Optional[Tensor]:
    """Ancestral sampling of `total` images over all ranks: the chains are independent, so rank r runs
    sample_fn(n_r, first_r) -> [n_r, C, H, W] for its contiguous share [first_r, first_r + n_r) (shard_range) with no
    inter-GPU traffic during the T steps, and ONE gather at the end brings the images to rank 0 (returned there in global
    order; None elsewhere).  The reference samples on rank 0 only (base_trainer.py:146-147,337-349).

    sample_fn receives the global index of its first image so that it can key its noise by global image index: a
    graphs.GraphedSampler built with `seed=job_seed, first_image=first_r` draws the same numbers for image i whatever the
    number of ranks, which makes the gathered result independent of the world size."""
    world = td.get_world_size(group) if td.is_initialized() else 1
    rank = td.get_rank(group) if td.is_initialized() else 0
    b, e = shard_range(total, rank, world)
    x = sample_fn(e - b, b) if e > b else None
    if world == 1:
        return x
    # every rank must take part in the gather with equally shaped buffers: pad the shares to the largest one
    most = (total + world - 1) // world
    meta = torch.zeros(4, dtype=torch.long, device=x.device if x is not None else "cpu")
    if x is not None:
        meta.copy_(torch.tensor(x.shape, dtype=torch.long))
    shapes = [torch.zeros_like(meta) for _ in range(world)]
    td.all_gather(shapes, meta, group=group)
    ref = next(s for s in shapes if s[0] > 0)
    chw = tuple(int(v) for v in ref[1:])
    dev = meta.device
    buf = torch.zeros((most,) + chw, device=dev, dtype=torch.float32)
    if x is not None:
        buf[:e - b].copy_(x)
    parts = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    td.gather(buf, parts, dst=0, group=group)
    if rank != 0:
        return None
    out = []
    for r, part in enumerate(parts):
        rb, re_ = shard_range(total, r, world)
        out.append(part[:re_ - rb])
    return torch.cat(out, dim=0)
