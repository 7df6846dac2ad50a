"""OverlappedTrainStep's exchange bookkeeping on two CPU ranks (gloo) with a stand-in engine: which half of the flat
gradient buffer is handed to the exchange after which backward phase, in which order the exchanges start and are
waited for, what the first pass of a shape does (no half is final early), and that a micro-step which only accumulates
exchanges nothing.  The reference gets this behaviour from torch's DistributedDataParallel (base_trainer.py:95-100,
215-235 there: `no_sync` on accumulating micro-steps, bucketed all-reduce during backward on the others); the device
side of the same class is covered on the GPU by the world-1 RCCL tests."""
import socket

import torch
import torch.distributed as td
import torch.multiprocessing as mp

import dm_amd_bootstrap  # noqa: F401
from dm_amd.runtime.dp import GradReducer
from dm_amd.runtime.graphs import OverlappedTrainStep

K, MID, CUT, TOTAL = 4, 1, 40, 100           # phases, the phase that completes the decoder half, the halves' boundary
LATE = [(8, 24), (28, 40)]                   # the part of the encoder half that is final one phase before the end (two runs)
REST = [(0, 8), (24, 28)]


class _Grads:
    """flat_t: the torch-layout gradient buffer; the decoder half is [CUT, TOTAL), the encoder half [0, CUT)"""

    def __init__(self, rank):
        self.flat_t = torch.zeros(TOTAL)
        self.rank, self.log = rank, []

    def unpack_phase(self, prefix, which):
        assert prefix == "decoder."
        lo, hi = (CUT, TOTAL) if which == 0 else (0, CUT)
        # the values a rank contributes: (index + 1) * (rank + 1), written when the half becomes final
        self.flat_t[lo:hi] = (torch.arange(lo, hi, dtype=torch.float32) + 1) * (self.rank + 1)
        self.log.append(("unpack", which))
        return lo, hi

    def unpack_where(self, tag, keep):
        ranges = LATE if tag == "encoder late" else REST
        for lo, hi in ranges:
            self.flat_t[lo:hi] = (torch.arange(lo, hi, dtype=torch.float32) + 1) * (self.rank + 1)
        self.log.append(("unpack", tag))
        return list(ranges)

    def torch_views(self):
        return {"encoder.w": self.flat_t[:CUT], "decoder.w": self.flat_t[CUT:]}


class _Engine:
    def __init__(self, rank, ready_at_mid):
        self.grads, self.ready_at_mid = _Grads(rank), ready_at_mid

    def backward_flush(self, ctx, p):
        return self.ready_at_mid if p == MID else True


class _UNet:
    _param_names = ["encoder.w", "decoder.w"]

    def __init__(self):
        self.ps = [torch.nn.Parameter(torch.zeros(CUT)), torch.nn.Parameter(torch.zeros(TOTAL - CUT))]

    def parameters(self):
        return iter(self.ps)


class _LoggingReducer(GradReducer):
    def __init__(self, *a, log, **kw):
        super().__init__(*a, **kw)
        self.log = log

    def start_range(self, flat, lo, hi):
        self.log.append(("start", lo, hi))
        return super().start_range(flat, lo, hi) + ((lo, hi),)

    def finish(self, handle):
        self.log.append(("finish",) + handle[-1])
        super().finish(handle[:-1])


def _drive(rank, ready_at_mid, final, enc_split=None):
    """one step of the driver's host logic: per phase the leaf work, then the exchange hook; then the wait"""
    step = object.__new__(OverlappedTrainStep)        # the constructor needs a GPU (streams, capture): wire the host state by hand
    step.eng, step.unet = _Engine(rank, ready_at_mid), _UNet()
    step.K, step.mid = K, MID
    step._enc_split, step._late_done, step._enc_late, step._enc_rest = enc_split, False, [], []
    step._ctx, step._ranges, step._ready, step._bound, step._handles = object(), [None, None], [False] * K, False, []
    step._sep_adam, step._final = True, final
    log = step.eng.grads.log
    step.reducer = _LoggingReducer(step.unet.ps, log=log, bucket_bytes=64)      # 16-element pieces: several collectives per range
    for p in range(K):
        step._leaf(p)
        step._exchange(p)
    step._finish_exchange()
    return step, log


def _worker(rank, world, port, q):
    td.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        mean = (torch.arange(TOTAL, dtype=torch.float32) + 1) * (1 + world) / 2
        own = (torch.arange(TOTAL, dtype=torch.float32) + 1) * (rank + 1)
        res = {}
        # steady state: the decoder half is final after phase MID and travels while phases MID+1.. compute
        step, log = _drive(rank, True, True)
        res["steady_log"] = log == [("unpack", 0), ("start", CUT, TOTAL), ("unpack", 1), ("start", 0, CUT),
                                    ("finish", CUT, TOTAL), ("finish", 0, CUT)]
        res["steady_mean"] = torch.allclose(step.eng.grads.flat_t, mean)
        res["bound"] = all(p.grad is not None and p.grad.data_ptr() == v.data_ptr()
                           for p, v in zip(step.unet.ps, (step.eng.grads.flat_t[:CUT], step.eng.grads.flat_t[CUT:])))
        res["ctx_released"] = step._ctx is None and step._handles == []
        # first pass of a shape: nothing is final before the last phase; both halves start there, decoder first
        step, log = _drive(rank, False, True)
        res["first_log"] = log == [("unpack", 0), ("unpack", 1), ("start", CUT, TOTAL), ("start", 0, CUT),
                                   ("finish", CUT, TOTAL), ("finish", 0, CUT)]
        res["first_mean"] = torch.allclose(step.eng.grads.flat_t, mean)
        # a micro-step that only accumulates: the halves are still unpacked, nothing is exchanged
        step, log = _drive(rank, True, False)
        res["accum_log"] = log == [("unpack", 0), ("unpack", 1)]
        res["accum_local"] = torch.equal(step.eng.grads.flat_t, own)
        # the encoder half in two parts: what is final after the second-to-last phase starts there, the small rest at the end
        step, log = _drive(rank, True, True, enc_split=1)
        want = ([("unpack", 0), ("start", CUT, TOTAL), ("unpack", "encoder late")] + [("start",) + r for r in LATE] +
                [("unpack", "encoder rest")] + [("start",) + r for r in REST] +
                [("finish", CUT, TOTAL)] + [("finish",) + r for r in LATE] + [("finish",) + r for r in REST])
        res["split_log"] = log == want
        res["split_mean"] = torch.allclose(step.eng.grads.flat_t, mean)
        # first pass of a shape with the split on: nothing is final early, the whole encoder half goes out at the end
        step, log = _drive(rank, False, True, enc_split=1)
        res["split_first_log"] = log == [("unpack", 0), ("unpack", 1), ("start", CUT, TOTAL), ("start", 0, CUT),
                                         ("finish", CUT, TOTAL), ("finish", 0, CUT)]
        res["split_first_mean"] = torch.allclose(step.eng.grads.flat_t, mean)
        step, log = _drive(rank, True, False, enc_split=1)
        res["split_accum_log"] = log == [("unpack", 0), ("unpack", "encoder late"), ("unpack", "encoder rest")]
        q.put((rank, res))
    finally:
        td.destroy_process_group()


def test_overlapped_step_exchanges_the_right_ranges_in_order():
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, r in res:
        bad = [k for k, ok in r.items() if not ok]
        assert not bad, f"rank {rank}: {bad}"
