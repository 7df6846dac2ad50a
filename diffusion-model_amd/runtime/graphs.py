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

    def replay(self) -> Tensor:
        if self.graph is None:
            self._body()
        else:
            self.graph.replay()
            if self.graph2 is not None:
                self.after_backward()
                self.graph2.replay()
        return self.loss


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
