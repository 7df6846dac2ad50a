"""HIP-graph capture of the two hot loops (through torch.cuda.graphs, which captures the stream the HIP kernels are
launched on): one training micro-step + optimizer step, and one ancestral sampling step replayed T times.

The reference drives both loops from Python with one kernel launch per torch op (base_trainer.py:196-235,
ddpm.py:153-162 there); at sub-millisecond step times the host cannot keep up, so the whole step is recorded
once and replayed.  Everything that must change between replays lives in device memory: the Philox seeds are
advanced by a kernel, timesteps are drawn by the captured torch RNG, `times -= 1` is a captured kernel."""
from __future__ import annotations

from typing import Callable, Optional

import torch
from torch import Tensor


def _engines(ddpm):
    core = ddpm.module if hasattr(ddpm, "module") else ddpm
    return [core.model._get_engine()]


class GraphedTrainStep:
    """randint t -> q -> UNet -> ELBO -> backward -> [grad hook, e.g. all-reduce] -> optimizer.step, as one graph.

    `x0` is a static input buffer: copy a new batch into `self.x0` before each `replay()`.
    The optimizer must be capturable (FusedAdam, or torch.optim.Adam(..., capturable=True)).

    With a gradient hook (data parallel) the step is captured as TWO graphs - everything up to the end of backward,
    and the optimizer step - and the hook runs eagerly between them: the collective is one launch per step, and
    keeping it out of the capture means the step does not depend on the collective library's graph support."""

    def __init__(self, ddpm, criterion, optimizer, x0: Tensor, *, after_backward: Optional[Callable[[], None]] = None,
                 use_graph: bool = True, warmup: int = 3):
        self.ddpm, self.criterion, self.optimizer = ddpm, criterion, optimizer
        self.core = ddpm.module if hasattr(ddpm, "module") else ddpm
        self.x0 = x0
        self.after_backward = after_backward
        self.loss = torch.zeros((), device=x0.device)
        self.graph = None
        for e in _engines(ddpm):
            e.always_repack = True
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream().wait_stream(s)
        self.graph2 = None
        if use_graph:
            self.optimizer.zero_grad(set_to_none=True)
            # "thread_local": helper threads of the process (the collective library's watchdog polling its events) must not
            # invalidate the capture; kernels launched by the autograd thread are captured all the same (capture is per stream)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self._front()
                if self.after_backward is None:
                    self._back()
            if self.after_backward is not None:
                self.after_backward()
                torch.cuda.synchronize()
                self.graph2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph2, capture_error_mode="thread_local"):
                    self._back()

    def _front(self):
        n = self.x0.shape[0]
        self.optimizer.zero_grad(set_to_none=True)
        t = torch.randint(1, self.core.T + 1, (n,), device=self.x0.device)
        x_t = self.core.q(self.x0, t)
        pred = self.ddpm(x_t, t)
        loss = self.criterion(self.ddpm, t, pred, self.x0, is_ddp=hasattr(self.ddpm, "module"))["Loss"]
        loss.backward()
        self.loss.copy_(loss.detach())

    def _back(self):
        self.optimizer.step()

    def _body(self):
        self._front()
        if self.after_backward is not None:
            self.after_backward()
        self._back()

    def local_body(self):
        """one eager step without the gradient hook (per-kernel timing on one rank must not enter a collective)"""
        self._front()
        self._back()

    def replay(self) -> Tensor:
        if self.graph is None:
            self._body()
        else:
            self.graph.replay()
            if self.graph2 is not None:
                self.after_backward()
                self.graph2.replay()
        return self.loss


class OverlappedTrainStep:
    """The same training step for data-parallel runs, with the gradient exchange overlapped with the backward pass.

    The reference's DDP wrapper overlaps bucketed all-reduces with autograd's backward (base_trainer.py:95-100 there).  Here
    the backward pass is the engine's tape, split in two phases, and the step is three HIP graphs with the collectives
    launched between them on the communicator's stream:

        graph A: randint t -> q -> UNet forward -> ELBO + its gradient -> backward phase 0 (head, decoder)
                 -> decoder gradients to torch layout            | all-reduce(decoder half) starts
        graph B: backward phase 1 (encoder) -> encoder gradients | ... and runs under graph B; all-reduce(encoder half)
        graph C: wait for both -> fused Adam step

    The step drives the engine directly (no autograd node), so it needs this package's DDPM/BaseUNet, ELBOLoss and
    FusedAdam.  Parameter .grad tensors are bound once to views of the engine's persistent flat gradient buffer."""

    def __init__(self, ddpm, criterion, optimizer, x0: Tensor, reducer, *, use_graph: bool = True, warmup: int = 3):
        self.core = ddpm.module if hasattr(ddpm, "module") else ddpm
        self.criterion, self.optimizer, self.reducer, self.x0 = criterion, optimizer, reducer, x0
        self.unet = self.core.model
        self.eng = self.unet._get_engine()
        self.eng.always_repack = True
        self.eng.bucketed = True
        self.loss = torch.zeros(1, device=x0.device)
        self._ctx = None
        self._ranges = [None, None]
        self._bound = False
        self.graphs = None
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream().wait_stream(s)
        if use_graph:
            torch.cuda.synchronize()
            gs = []
            for fn, between in ((self._phase_a, lambda: self._exchange(0)), (self._phase_b, lambda: self._exchange(1)),
                                (self._phase_c, None)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    fn()
                gs.append(g)
                if between is not None:
                    between()
                else:
                    torch.cuda.synchronize()
            self.graphs = gs

    # ---- the three phases --------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _phase_a(self):
        from . import hip
        core, n = self.core, self.x0.shape[0]
        t = torch.randint(1, core.T + 1, (n,), device=self.x0.device)
        x_t = core.q(self.x0, t)
        params = dict(zip(self.unet._param_names, self.unet.parameters()))
        pred, self._ctx = self.eng.forward(params, x_t, t, train=self.unet.training, record=True)
        table = self.criterion._table(core, pred.device)
        chw = pred.numel() // n
        flags = (int(self.criterion.bce), int(bool(self.criterion.use_elbo)))
        hip.call("ddpm_elbo_fwd", hip.ptr(pred), hip.ptr(self.x0), hip.ptr(t), hip.ptr(table), hip.ptr(self.loss), n, chw, *flags, hip.stream())
        dpred = torch.empty_like(pred)
        hip.call("ddpm_elbo_bwd", hip.ptr(pred), hip.ptr(self.x0), hip.ptr(t), hip.ptr(table), None, hip.ptr(dpred), n, chw, *flags, hip.stream())
        self._ready0 = self.eng.backward_phase(self._ctx, 0, dpred)
        if self._ready0:
            self._ranges[0] = self.eng.grads.unpack_phase("decoder.", 0)

    @torch.no_grad()
    def _phase_b(self):
        self.eng.backward_phase(self._ctx, 1)
        self._ctx = None
        if not self._ready0:                       # first pass of a shape: everything becomes final here
            self._ranges[0] = self.eng.grads.unpack_phase("decoder.", 0)
        self._ranges[1] = self.eng.grads.unpack_phase("decoder.", 1)
        if not self._bound:
            views = self.eng.grads.torch_views()
            for name, p in zip(self.unet._param_names, self.unet.parameters()):
                p.grad = views[name]
            self._bound = True

    @torch.no_grad()
    def _phase_c(self):
        self.optimizer.step()

    def _exchange(self, k: int):
        """start the exchange of half k (0: decoder, 1: encoder; after phase B also finish both)"""
        flat = self.eng.grads.flat_t
        if k == 0:
            self._h0 = self.reducer.start_range(flat, *self._ranges[0]) if self._ready0 else None
            return
        if not self._ready0:
            self._h0 = self.reducer.start_range(flat, *self._ranges[0])
        h1 = self.reducer.start_range(flat, *self._ranges[1])
        self.reducer.finish(self._h0)
        self.reducer.finish(h1)

    def _body(self):
        self._phase_a()
        self._exchange(0)
        self._phase_b()
        self._exchange(1)
        self._phase_c()

    def local_body(self):
        """one eager step without the exchange (per-kernel timing on one rank must not enter a collective)"""
        self._phase_a()
        self._phase_b()
        self._phase_c()

    def replay(self) -> Tensor:
        if self.graphs is None:
            self._body()
        else:
            a, b, c = self.graphs
            a.replay()
            self._exchange(0)
            b.replay()
            self._exchange(1)
            c.replay()
        return self.loss[0]


class GraphedSampler:
    """One sampling step (UNet forward + reverse step + t -= 1) captured once and replayed T times per chain."""

    def __init__(self, ddpm, n: int, shape, *, use_graph: bool = True):
        from torch.nn.modules.utils import _pair
        self.ddpm = ddpm
        dev = next(ddpm.parameters()).device
        self.x = torch.zeros((n, ddpm.img_channels) + _pair(shape), device=dev)
        self.x_next = torch.empty_like(self.x)
        self.times = torch.zeros(n, device=dev, dtype=torch.long)
        self.graph = None
        self.bce = ddpm.reconstruction_method == "bce"
        self.times.fill_(ddpm.T)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            self._body()
        torch.cuda.current_stream().wait_stream(s)
        if use_graph:
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"), torch.no_grad():
                self._body()

    def _body(self):
        x0_pred = self.ddpm(self.x, self.times)
        self.ddpm.reverse_q(self.x, x0_pred, self.times, out=self.x_next, _sigmoid=self.bce)
        self.x.copy_(self.x_next)
        self.times -= 1

    @torch.no_grad()
    def sample(self, x_T: Optional[Tensor] = None) -> Tensor:
        """Run the full T-step chain; returns the final tensor (before the (x+1)/2 rescale)."""
        if x_T is None:
            self.x.normal_()
        else:
            self.x.copy_(x_T)
        self.times.fill_(self.ddpm.T)
        for _ in range(self.ddpm.T):
            if self.graph is not None:
                self.graph.replay()
            else:
                self._body()
        return self.x
