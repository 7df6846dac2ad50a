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


_STREAM_POOL: dict = {}


def _pool_stream(device, k: int = 0) -> "torch.cuda.Stream":
    """Stream k of a small per-device pool created once, in one go, for the life of the process: the side stream of the
    pipelined drivers (k = 0) and the streams of concurrent sampling chains.  Streams are dealt onto a few hardware queues in
    creation order, so a fresh stream per driver instance can land on the queue the main stream uses, and the two halves of
    the pipeline then run one after the other (measured: 7.15 instead of 6.47 ms per 64x64x3 step for the second driver of a
    process).  The first streams a process creates sit on queues of their own."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _STREAM_POOL:
        _STREAM_POOL[key] = [torch.cuda.Stream(device=key) for _ in range(4)]
    pool = _STREAM_POOL[key]
    while k >= len(pool):
        pool.append(torch.cuda.Stream(device=key))
    return pool[k]


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
        t = self.core.sample_timesteps(n, self.x0.device)
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
        if hasattr(self.optimizer, "sync_lr"):
            self.optimizer.sync_lr()            # an LR scheduler's new value reaches the captured Adam through device memory
        if self.graph is None:
            self._body()
        else:
            self.graph.replay()
            if self.graph2 is not None:
                self.after_backward()
                self.graph2.replay()
        return self.loss


def _merge_ranges(ranges):
    """sorted union of half-open element ranges, touching ones joined"""
    out = []
    for lo, hi in sorted(ranges):
        if out and lo <= out[-1][1]:
            out[-1] = (out[-1][0], max(out[-1][1], hi))
        else:
            out.append((lo, hi))
    return out


class PipelinedTrainStep:
    """The training step as a small pipeline of HIP graphs on two streams, driving the engine directly (no autograd node).

    The backward pass is latency bound: its data-gradient chain is a long sequence of small dependent kernels that leaves
    most of the machine idle, while the weight gradients, the partial-sum folds and the time-embedding MLPs are leaves that
    nothing in the chain waits for.  The engine therefore runs the backward in `phases` pieces (head-first), and this
    driver replays, per phase p, a graph M_p (data-gradient chain) on the main stream and a graph W_p (the phase's leaf
    work) on a side stream, W_p beside M_{p+1}:

        main:  [A: t, q, forward, ELBO, M_0] [M_1] [M_2] [M_3] ........ wait side ( -> [C: Adam] when data parallel)
        side:                                 [W_0]  [W_1]  [W_2]  [W_3 + gradients to torch layout + Adam + weight
                                                                    re-layout] [next step's dropout bits]

    Single GPU: the optimizer step closes the last leaf graph (every graph start costs tens of microseconds on the device,
    so the critical path holds as few as possible), followed by the weight re-layout; the next step's dropout keep bits
    (double-buffered in the engine: mask_ahead) are two plain launches behind it that the main stream does not wait for:
    they run inside the start-up latency of the next step's first graph.

    (4 phases measured best: with one phase per stage the leaf launches become too small to keep the side stream ahead.
    The last phase's leaf work runs alone, so its weight-gradient problems are cut into more, shorter position splits (4 instead of 8 tiles each).)

    Data parallel (reducer given): the reference's DDP wrapper overlaps bucketed all-reduces with autograd's backward
    (base_trainer.py:95-100 there).  Here the `decoder.*` half of the flat gradient buffer is final after the middle
    phase: it is converted to torch layout there and its all-reduce (RCCL, AVG) starts at once, travelling while the
    encoder's phases compute; the `encoder.*` half follows after the last phase.  Collectives are launched eagerly
    between graph replays, so nothing depends on capturing the collective library.

    Needs this package's DDPM/BaseUNet, ELBOLoss and FusedAdam.  Parameter .grad tensors are bound once to views of the
    engine's persistent flat gradient buffer.  Copy a new batch into `self.x0` before each replay()."""

    def __init__(self, ddpm, criterion, optimizer, x0: Tensor, reducer=None, *, phases: int = 4, use_graph: bool = True, warmup: int = 3,
                 gradient_accumulation_steps=1, t: Optional[Tensor] = None, noise: Optional[Tensor] = None, side_stream=None):
        """gradient_accumulation_steps = k > 1 (the reference trainer's default is 3, base_trainer.py:43,215-235 there):
        gradients of k consecutive replays are summed without any 1/k scaling, the optimizer steps on every k-th replay
        (or on replay(last=True), the trainer's "last batch of the epoch" case) and the sum restarts after it; with a
        reducer the exchange happens once, on the summed gradients (same numbers as the reference's exchange per
        micro-batch, a third of the traffic).  The value may arrive as a string (CLI passthrough).
        t / noise: static device tensors (int64 [N] / like x0) that replace the in-kernel draws of the per-image timestep
        and of the forward-noising noise, for parity checks; overwrite them between replays."""
        self.core = ddpm.module if hasattr(ddpm, "module") else ddpm
        self.criterion, self.optimizer, self.reducer, self.x0 = criterion, optimizer, reducer, x0
        self.t_in, self.noise_in = t, noise
        self.accum = max(1, int(gradient_accumulation_steps))
        self._sep_adam = reducer is not None or self.accum > 1      # the optimizer step is a graph of its own on the main stream
        self._micro = 0
        self.unet = self.core.model
        self.eng = self.unet._get_engine()
        # the kernel-layout weight copies are refreshed right after the optimizer step (see _adam), not at the head of the
        # next forward pass: there the launch falls into the start-up latency of the next step's first graph
        self.eng.always_repack = False
        self.eng.mask_ahead = True
        if int(phases) < 2:
            raise ValueError("phases must be 2, 4 or larger (= one phase per stage)")
        self.eng.phases = int(phases)
        self.eng.external_zero = self.accum > 1     # the driver clears the gradient sums (every k-th replay), not the backward pass
        import os
        self.eng.wgrad_tiles_last = int(os.environ.get("DDPM_TPS_LAST", self.eng.wgrad_tiles_last))
        self.eng.wgrad_tiles_per_split = int(os.environ.get("DDPM_TPS", self.eng.wgrad_tiles_per_split))
        if os.environ.get("DDPM_TPS_PHASES"):
            self.eng.wgrad_tiles_by_phase = [int(v) for v in os.environ["DDPM_TPS_PHASES"].split(",")]
        # data parallel with the standard four phases (cuts: before encoder stage e, before the decoder's input, before decoder
        # stage d): the encoder's half of the gradients is exchanged in two parts, see _leaf
        self._enc_split, self._late_done, self._enc_late, self._enc_rest = None, False, [], []
        self._early, self._early_late, self._early_done = False, False, 0
        cuts = self.eng._phase_cuts(len(self.eng.spec.stages))
        enc_cuts = [c for c in cuts if c.startswith("enc") and c[3:].isdigit()]
        four = len(cuts) == 3 and len(enc_cuts) == 1 and "dec" in cuts
        if reducer is not None and four and os.environ.get("DDPM_DP_ENC_SPLIT", "1") != "0":
            self._enc_split = int(enc_cuts[0][3:])
        # single GPU with the standard four phases: the optimizer step and the weight re-layout of the decoder's parameters are
        # taken as soon as their gradients are final (after the middle phase, where the side stream has slack), the rest at
        # the end of the step, which nothing hides: half the Adam launch and half the re-layout leave the step's tail
        # (2.707 -> 2.695 ms at 32x32x1, 4.99 -> 4.96 ms at 64x64x3).  DDPM_EARLY_ADAM=2 also updates the deep encoder stages
        # one phase before the end: measured slower, that phase's leaf work is what the last phase's leaf work waits for.
        self._early = not self._sep_adam and four and os.environ.get("DDPM_EARLY_ADAM", "1") != "0"
        self._early_late = self._early and os.environ.get("DDPM_EARLY_ADAM", "1") == "2"
        if self._early:
            # (the gradients of the deep encoder stages go to torch layout one phase before the end either way: that launch is
            # then not part of the step's tail)
            self._enc_split = int(enc_cuts[0][3:])
        self._fuse_head = os.environ.get("DDPM_FUSE_HEAD", "1") != "0"
        self.K, self.mid = self.eng.phase_layout()      # mid: the phase after which the decoder half of the gradients is final
        self.loss = torch.zeros(1, device=x0.device)
        self._head_done = torch.zeros(1, device=x0.device, dtype=torch.int32)      # ddpm_train_head's last-workgroup counter
        self._ctx = None
        self._ranges = [None, None]
        self._ready = [False] * self.K
        self._bound = False
        self._handles = []
        self.graphs = None
        self.side = side_stream if side_stream is not None else _pool_stream(x0.device, 0)
        self.events = [torch.cuda.Event() for _ in range(self.K)]
        self.ev_done = torch.cuda.Event()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._run(None)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if use_graph:
            # capture in execution order; every graph has its own memory pool, so what a side graph still reads is never
            # handed to a later main graph's allocations
            gm, gw = [], []
            gp = self.eng._mask_key is not None         # dropout: the next step's keep bits are drawn beside the forward pass
            for p in range(self.K):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._main(p)
                gm.append(g)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self.side, capture_error_mode="thread_local"):
                    self._leaf(p)
                gw.append(g)
            gc = None
            if self._sep_adam:
                gc = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gc, capture_error_mode="thread_local"):
                    self._adam()
            torch.cuda.synchronize()
            self.graphs = (gm, gw, gc, gp)

    # ---- pieces ------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _main(self, p: int):
        """main-stream work of phase p (phase 0 also holds the whole forward pass)"""
        if p > 0:
            self.eng.backward_data(self._ctx, p)
            return
        from . import hip
        core, n = self.core, self.x0.shape[0]
        params = dict(zip(self.unet._param_names, self.unet.parameters()))
        # timestep draw, forward noising (straight into the stem convolution's input layout), time embedding and seed advance
        # in one launch: the same numbers as core.sample_timesteps + core.q + the engine's own layout bridge and embedding
        _, c, h, w = self.x0.shape
        dev = self.x0.device
        sp = self.eng.spec
        cpad = (c + 15) // 16 * 16
        t = torch.empty(n, device=dev, dtype=torch.long)
        xin = torch.empty(n, h, w, cpad, device=dev, dtype=torch.bfloat16)
        sin = torch.empty(n, sp.t_dim, device=dev, dtype=torch.bfloat16)
        x0 = self.x0 if self.x0.dtype == torch.float32 and self.x0.is_contiguous() else self.x0.float().contiguous()
        hip.call("ddpm_train_head", hip.ptr(x0), hip.ptr(self.t_in), hip.ptr(self.noise_in), hip.ptr(core.schedule_table(dev)), hip.ptr(t),
                 hip.ptr(xin), cpad, hip.ptr(sin), sp.time_norm, sp.t_dim, n, c, h * w, core.T, hip.ptr(core._rng_seed(dev)),
                 hip.ptr(self._head_done), hip.stream())
        pred, self._ctx = self.eng.forward(params, None, t, train=self.unet.training, record=True, mask_ahead=True, head=(xin, sin))
        table = self.criterion._table(core, pred.device)
        chw = pred.numel() // n
        flags = (int(self.criterion.bce), int(bool(self.criterion.use_elbo)))
        put_dz, bias_grad, (_, hh, ww, cp, ci) = self.eng.head_gradient_slot()
        if self._fuse_head and not self.criterion.bce and cp == 16 and self.x0.dtype == torch.float32 and self.x0.is_contiguous():
            # squared-error loss of the tanh-headed network: loss, its gradient, the tanh backward into the last convolution's
            # layout and that convolution's bias gradient in one launch (bit-identical gradient; four launches otherwise)
            self.eng.backward_begin()
            dz = torch.empty(n, hh, ww, cp, device=pred.device, dtype=torch.bfloat16)
            hip.call("ddpm_elbo_tanh_head", hip.ptr(pred), hip.ptr(self.x0), hip.ptr(t), hip.ptr(table), hip.ptr(self.loss), hip.ptr(dz),
                     hip.ptr(bias_grad), n, ci, hh * ww, flags[1], hip.stream())
            put_dz(dz)
            self.eng.backward_data(self._ctx, 0, None, begun=True)
            return
        hip.call("ddpm_elbo_fwd", hip.ptr(pred), hip.ptr(self.x0), hip.ptr(t), hip.ptr(table), hip.ptr(self.loss), n, chw, *flags, hip.stream())
        dpred = torch.empty_like(pred)
        hip.call("ddpm_elbo_bwd", hip.ptr(pred), hip.ptr(self.x0), hip.ptr(t), hip.ptr(table), None, hip.ptr(dpred), n, chw, *flags, hip.stream())
        self.eng.backward_data(self._ctx, 0, dpred)

    @torch.no_grad()
    def _leaf(self, p: int):
        """side-stream work of phase p: its weight gradients and folds; the halves of the gradient buffer go to torch layout
        as soon as they are final"""
        g = self.eng.grads
        self._ready[p] = self.eng.backward_flush(self._ctx, p)
        last = p == self.K - 1
        if last:
            self._ctx = None
        if p == 0:
            self._late_done = False
        early = getattr(self, "_early", False) and self._bound       # (the first pass binds the .grad views the optimizer steps on)
        if p == 0:
            self._early_done = 0
        if p == self.mid and self._ready[p]:
            self._ranges[0] = g.unpack_phase("decoder.", 0)
            if early:
                self._adam_piece([self._ranges[0]], "decoder", lambda n: n.startswith("decoder."), False)
        if self._enc_split is not None and p == self.K - 2 and self._ready[p] and self._ready[self.mid]:
            # the encoder's deep stages and the latent layers (most of the encoder's bytes) are final one phase before the
            # end.  Data parallel: their exchange travels during the last phase, which leaves a few hundred KB for the end
            self._enc_late = g.unpack_where("encoder late", self._is_enc_late)
            self._late_done = True
            if early and self._early_done == 1 and self._early_late:
                self._adam_piece(self._enc_late, "encoder late", self._is_enc_late, False)
        if last:
            if not self._ready[self.mid]:               # first pass of a shape: everything becomes final here
                self._ranges[0] = g.unpack_phase("decoder.", 0)
            if self._late_done:
                self._enc_rest = g.unpack_where("encoder rest", self._is_enc_rest)
            else:
                self._ranges[1] = g.unpack_phase("decoder.", 1)
            if not self._bound:
                views = g.torch_views()
                for name, prm in zip(self.unet._param_names, self.unet.parameters()):
                    prm.grad = views[name]
                self._bound = True
            if self._early_done == 2:
                self._adam_piece(self._enc_rest, "encoder rest", self._is_enc_rest, True)
                self.eng.commit_masks()
            elif self._early_done == 1:
                rest = _merge_ranges(self._enc_late + self._enc_rest) if self._late_done else [self._ranges[1]]
                self._adam_piece(rest, "not decoder", lambda n: not n.startswith("decoder."), True)
                self.eng.commit_masks()
            elif not self._sep_adam:
                # single GPU: nothing comes between the last gradient and the optimizer, so the update rides at the end
                # of this graph instead of paying a graph start of its own on the main stream
                self._adam()

    def _is_enc_late(self, name: str) -> bool:
        """a parameter of the encoder whose gradient is final after the second-to-last backward phase: encoder stages
        >= self._enc_split with their time-embedding MLPs, and the latent layers"""
        for kind in ("encoder.layers.", "encoder.t_embeddings."):
            if name.startswith(kind):
                return int(name[len(kind):].split(".", 1)[0]) >= self._enc_split
        return name.startswith("encoder.latent_layers.")

    def _is_enc_rest(self, name: str) -> bool:
        return not name.startswith("decoder.") and not self._is_enc_late(name)

    @torch.no_grad()
    def _adam_piece(self, ranges, tag: str, keep, last: bool):
        """the optimizer step and the weight re-layout for the parameters whose gradients just became final: the element ranges
        `ranges` of the flat buffers, the layers whose name passes `keep`.  No later phase of this step reads these weights:
        the data-gradient chain has passed their layers, and weight gradients read activations only."""
        self.optimizer.step_ranges(ranges, last)
        params = dict(zip(self.unet._param_names, self.unet.parameters()))
        self.eng.wc.refresh_where(tag, keep, params, self.x0.device, done=last)
        self._early_done += 1

    @torch.no_grad()
    def _adam(self):
        self.optimizer.step()
        self.eng.commit_masks()
        params = dict(zip(self.unet._param_names, self.unet.parameters()))
        self.eng.wc.refresh(params, self.x0.device, force=True)

    @torch.no_grad()
    def _pre(self):
        """side-stream work beside the forward pass: the NEXT step's dropout keep bits (ALU-bound, fits the gaps of the
        latency-bound forward), into the stand-by buffer that _adam() makes live after this step's last reader"""
        self.eng.generate_masks(ahead=True)

    def _exchange(self, p: int):
        """called with the side stream current, right after W_p was queued on it"""
        if self.reducer is None or not self._final:
            return
        flat = self.eng.grads.flat_t
        last = p == self.K - 1
        if p == self.mid and self._ready[p]:
            self._handles.append(self.reducer.start_range(flat, *self._ranges[0]))
        if self._late_done and p == self.K - 2:
            self._handles += [self.reducer.start_range(flat, lo, hi) for lo, hi in self._enc_late]
        if last:
            if not self._ready[self.mid]:
                self._handles.append(self.reducer.start_range(flat, *self._ranges[0]))
            if self._late_done:
                self._handles += [self.reducer.start_range(flat, lo, hi) for lo, hi in self._enc_rest]
            else:
                self._handles.append(self.reducer.start_range(flat, *self._ranges[1]))

    def _finish_exchange(self):
        """wait for the started range exchanges in the order they were started (every rank starts them in the same order,
        which is what a collective library requires)"""
        for h in self._handles:
            self.reducer.finish(h)
        self._handles.clear()

    def _run(self, graphs, exchange: bool = True, last: bool = False):
        main = torch.cuda.current_stream()
        gm, gw, gc, gp = graphs if graphs is not None else (None, None, None, None)
        self._final = last or (self._micro + 1) % self.accum == 0
        if self.accum > 1 and self._micro % self.accum == 0:
            self.eng.grads_zero()                       # a plain fill ahead of the first graph: the sums restart here
        self._micro = 0 if self._final else self._micro + 1
        if self._sep_adam:
            # data parallel: the commit of the stand-by dropout bits is part of the Adam graph on the main stream; the next
            # bits are drawn beside the forward pass by two plain launches on the side stream (a graph of their own would
            # delay the start of the forward graph behind it)
            if graphs is None:
                self._pre()
            elif gp:
                self.side.wait_event(self.ev_done)
                with torch.cuda.stream(self.side):
                    self._pre()
        for p in range(self.K):
            if gm is None:
                self._main(p)
            else:
                gm[p].replay()
            if gw is None:
                # eager: everything on one stream (tensors handed from one stream to another would need the allocator's
                # cross-stream bookkeeping; the captured graphs have private pools instead)
                self._leaf(p)
                if exchange:
                    self._exchange(p)
                continue
            self.events[p].record(main)
            self.side.wait_event(self.events[p])
            with torch.cuda.stream(self.side):
                gw[p].replay()
                if exchange:
                    self._exchange(p)
        if not self._sep_adam:
            # the optimizer step was part of the last leaf graph.  The next step's dropout bits follow it on the side stream
            # as two plain launches AFTER the point the main stream joins: they run inside the start-up latency of the next
            # step's first graph without that graph waiting for them
            if gw is None:
                self._pre()
                return
            self.ev_done.record(self.side)
            main.wait_event(self.ev_done)
            if gp:
                with torch.cuda.stream(self.side):
                    self._pre()
            return
        main.wait_stream(self.side)
        self._finish_exchange()
        if not self._final:
            self.eng.commit_masks()                     # a micro-step that only accumulates: no update, fresh dropout bits
        elif gc is None:
            self._adam()
        else:
            gc.replay()
        if gc is not None:
            self.ev_done.record(main)

    def local_body(self):
        """one eager step without the exchange (per-kernel timing on one rank must not enter a collective)"""
        self._run(None, exchange=False)

    def replay(self, last: bool = False) -> Tensor:
        """One micro-step on the batch in self.x0; last=True forces the optimizer step (the trainer's last batch of an epoch)."""
        if hasattr(self.optimizer, "sync_lr"):
            self.optimizer.sync_lr()            # an LR scheduler's new value reaches the captured Adam through device memory
        self._run(self.graphs, last=last)
        return self.loss[0]


class OverlappedTrainStep(PipelinedTrainStep):
    """PipelinedTrainStep for data-parallel runs (the gradient exchange overlapped with the backward pass)."""

    def __init__(self, ddpm, criterion, optimizer, x0: Tensor, reducer, **kw):
        super().__init__(ddpm, criterion, optimizer, x0, reducer, **kw)


class ConcurrentSampler:
    """Several GraphedSamplers of `n` images each, their chains advanced in lockstep on separate streams.  Chains are
    independent and a single chain's kernels leave most of the machine idle at the low-resolution stages, so chains on
    different streams fill each other's gaps.  sample() returns the [streams * n, C, H, W] result."""

    def __init__(self, ddpm, n: int, shape, streams: int = 2, *, use_graph: bool = True, first_image: int = 0,
                 seed: Optional[int] = None):
        self.ddpm = ddpm
        dev = next(ddpm.parameters()).device
        self.streams = [_pool_stream(dev, k) for k in range(streams)]
        self.samplers = []
        for k, s in enumerate(self.streams):
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self.samplers.append(GraphedSampler(ddpm, n, shape, use_graph=use_graph, stream=s, own_engine=True,
                                                    first_image=first_image + k * n, seed=seed))
        for s in self.streams:
            torch.cuda.current_stream().wait_stream(s)

    @torch.no_grad()
    def sample(self) -> Tensor:
        main = torch.cuda.current_stream()
        for s, smp in zip(self.streams, self.samplers):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                smp.begin()
        for _ in range(self.ddpm.T):
            for s, smp in zip(self.streams, self.samplers):
                with torch.cuda.stream(s):
                    smp.step()
        for s in self.streams:
            main.wait_stream(s)
        return torch.cat([smp.x for smp in self.samplers], dim=0)


class GraphedSampler:
    """One sampling step (UNet forward + reverse step + t -= 1) captured once and replayed T times per chain.

    seed / first_image make a sampling job independent of how it is split: with a job seed, the start tensor and every
    step's noise are Philox draws keyed by (seed, chain number, step, GLOBAL element index), the sampler holding images
    first_image .. first_image + n - 1 of the job.  Two GPUs sampling images [0, n) and [n, 2n) with the same seed then draw
    exactly the numbers one GPU sampling [0, 2n) draws.  Without a seed the sampler owns a private random stream."""

    _CHAIN_STRIDE = 0x632BE59BD9B4E019            # seed distance between consecutive chains of one job

    def __init__(self, ddpm, n: int, shape, *, use_graph: bool = True, stream=None, own_engine: bool = True,
                 first_image: int = 0, seed: Optional[int] = None):
        from torch.nn.modules.utils import _pair
        self.ddpm = ddpm
        self.job_seed = None if seed is None else int(seed)
        self._chains = 0
        # The sampler drives an engine of its own: the kernel-layout weight copies are refreshed once per chain (begin)
        # instead of inside every captured step, and samplers on different streams do not share per-shape bookkeeping.
        self._engine = None
        if own_engine:
            from .engine import UNetEngine
            self._engine = UNetEngine(ddpm.model.spec)
        dev = next(ddpm.parameters()).device
        self.seed = torch.randint(0, 2 ** 62, (1,), device=dev, dtype=torch.long)     # this chain's noise stream
        self.x = torch.zeros((n, ddpm.img_channels) + _pair(shape), device=dev)
        self.first_index = int(first_image) * (self.x.numel() // n)
        if self.first_index % 4:
            raise ValueError("GraphedSampler: first_image * C * H * W must be a multiple of 4")
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
            with torch.cuda.graph(self.graph, stream=stream, capture_error_mode="thread_local"), torch.no_grad():
                self._body()

    def _forward(self):
        if self._engine is None:
            return self.ddpm(self.x, self.times)
        return self._engine.forward(self._params(), self.x, self.times, train=self.ddpm.model.training, record=False)[0]

    def _params(self):
        unet = self.ddpm.model
        return dict(zip(unet._param_names, unet.parameters()))

    def begin(self, x_T: Optional[Tensor] = None):
        """start a chain: x_T (or fresh noise), t = T, current weights"""
        if self.job_seed is not None:
            self.seed.fill_((self.job_seed + self._chains * self._CHAIN_STRIDE) & (2 ** 62 - 1))
            self._chains += 1
        if x_T is not None:
            self.x.copy_(x_T)
        elif self.job_seed is None:
            self.x.normal_()
        else:
            from . import hip
            hip.call("ddpm_fill_normal", hip.ptr(self.x), self.x.numel(), hip.ptr(self.seed), 1, self.first_index, hip.stream())
        self.times.fill_(self.ddpm.T)
        if self._engine is not None:
            self._engine.wc.refresh(self._params(), self.x.device, force=True)

    def step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self._body()

    def _body(self):
        x0_pred = self._forward()
        # in place: the step kernel reads and writes every element in the same thread (one copy per step less)
        self.ddpm.reverse_q(self.x, x0_pred, self.times, out=self.x, _sigmoid=self.bce, _seed=self.seed,
                            _first_index=self.first_index)
        self.times -= 1

    @torch.no_grad()
    def sample(self, x_T: Optional[Tensor] = None) -> Tensor:
        """Run the full T-step chain; returns the final tensor (before the (x+1)/2 rescale)."""
        self.begin(x_T)
        for _ in range(self.ddpm.T):
            self.step()
        return self.x
