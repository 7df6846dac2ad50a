"""BaseUNet: the denoiser backbone behind the reference's constructor and forward signature
(neural_nets/denoise_models/unets.py:243-273 there), executed by the HIP engine.

The module owns ordinary fp32 nn.Parameters with exactly the reference's state_dict keys (350 for the default 4-stage
network; swish / hardswish add one 0-dim `beta` per activation module), shapes and
registration order, so checkpoints, optimizers, EMA copies and DDP wrappers see the same object they would see
with the reference class.  The sub-modules are plain parameter containers: all arithmetic happens in
runtime/engine.py, entered through one autograd Function.
"""
from __future__ import annotations

import math
import weakref
from typing import Callable, Dict, List, Tuple

import torch
from torch import Tensor, nn

from ...runtime.engine import UNetEngine, UNetSpec
from ...utils.initializers import get_initializer
from ..activations import activation_code

# how a parameter is initialised (reference: unets.py:93-97,196-203; constituent_blocks.py:108-115,210-217,321-324;
# conv_block.py:121-129; torch defaults for everything not touched there)
W_INIT, W_INIT_EMBED_DEC, ONES, ZEROS, LINEAR_BIAS_DEFAULT = range(5)


class _Slot(nn.Module):
    """Parameter container addressed by dotted paths ('layers.0.layers.1.expansion.1.weight')."""

    def put(self, path: str, p: nn.Parameter) -> None:
        head, _, rest = path.partition(".")
        if not rest:
            self.register_parameter(head, p)
            return
        if head not in self._modules:
            self.add_module(head, _Slot())
        self._modules[head].put(rest, p)


def parameter_table(spec: UNetSpec) -> List[Tuple[str, Tuple[int, ...], int]]:
    """(name, shape, init rule) of every parameter in the reference's registration order."""
    out: List[Tuple[str, Tuple[int, ...], int]] = []
    st, te, ex = spec.stages, spec.t_dim, spec.expand

    def actp(site):
        """the activation module at `site`: swish / hardswish own a 0-dim `beta` = 1 (activations.py:40,60 there)"""
        if spec.act_scalar:
            out.append((site + ".beta", (), ONES))

    def conv(name, co, ci, k, bias_rule=ONES):
        out.append((name + ".weight", (co, ci, k, k), W_INIT))
        out.append((name + ".bias", (co,), bias_rule))

    def norm(name, c):
        out.append((name + ".weight", (c,), ONES))
        out.append((name + ".bias", (c,), ZEROS))

    def sir(pre, c):
        e = int(ex * c)
        conv(pre + "expansion.1", e, c, 1); norm(pre + "expansion.2", e); actp(pre + "expansion.3")
        conv(pre + "depthwise.1", e, 1, 3); norm(pre + "depthwise.2", e); actp(pre + "depthwise.3")
        conv(pre + "projection.1", c, e, 1); norm(pre + "projection.2", c)

    def embed(pre, c, rule):
        out.append((pre + "layers.0.weight", (2 * c, te), rule))
        out.append((pre + "layers.0.bias", (2 * c,), LINEAR_BIAS_DEFAULT))
        actp(pre + "layers.1")
        out.append((pre + "layers.2.weight", (c, 2 * c), rule))
        out.append((pre + "layers.2.bias", (c,), ONES))

    def attention(pre, c):
        # torch's default initialisers; the table's ZEROS differs from the block modules' constant, so map it here
        from .constituent_blocks import conditional_entries
        from .._block_base import ZEROS as BLOCK_ZEROS, ONES as BLOCK_ONES
        for name, shape, rule in conditional_entries(pre, c, spec.attn_depth, c, spec.act_name):
            out.append((name, shape, rule if callable(rule) else {BLOCK_ZEROS: ZEROS, BLOCK_ONES: ONES}[rule]))

    e = "encoder."
    conv(e + "stem_layers.0", st[0], spec.img_channels, 3); actp(e + "stem_layers.1")
    ins = [st[0]] + st[:-1]
    for i, (ci, co) in enumerate(zip(ins, st)):
        b = f"{e}layers.{i}.layers."
        conv(b + "0.0", co, ci, 3); norm(b + "0.1", co); actp(b + "0.2")
        k = 1
        if i in spec.attn_stages:
            attention(b + "1.", co)
            k = 2
        sir(f"{b}{k}.", co); sir(f"{b}{k + 1}.", co)
        conv(f"{b}{k + 2}.0", co, co, 3); norm(f"{b}{k + 2}.1", co); actp(f"{b}{k + 2}.2")
    for i, ci in enumerate(ins):
        embed(f"{e}t_embeddings.{i}.", ci, W_INIT)
    conv(e + "latent_layers.0", spec.latent_dim, st[-1], 3); norm(e + "latent_layers.1", spec.latent_dim); actp(e + "latent_layers.2")

    d = "decoder."
    rs = st[::-1]
    conv(d + "decoder_input_layers.0", rs[0], spec.latent_dim, 3); norm(d + "decoder_input_layers.1", rs[0]); actp(d + "decoder_input_layers.2")
    outs = rs[1:] + [rs[-1]]
    for i, (ci, co) in enumerate(zip(rs, outs)):
        b = f"{d}layers.{i}.layers."
        conv(b + "0.0", ci, 2 * ci, 3); norm(b + "0.1", ci); actp(b + "0.2")
        k = 1
        if len(st) - 1 - i in spec.attn_stages:
            attention(b + "1.", ci)
            k = 2
        sir(f"{b}{k}.", ci)
        out.append((f"{b}{k + 1}.0.weight", (ci, co, 2, 2), W_INIT)); out.append((f"{b}{k + 1}.0.bias", (co,), ONES)); norm(f"{b}{k + 1}.1", co)
        actp(f"{b}{k + 1}.2")
        sir(f"{b}{k + 2}.", co); sir(f"{b}{k + 3}.", co)
    for i, ci in enumerate(rs):
        embed(f"{d}t_embeddings.{i}.", ci, W_INIT_EMBED_DEC)
    conv(d + "out_layers.0", rs[-1], rs[-1], 3); actp(d + "out_layers.1")
    conv(d + "out_layers.2", spec.img_channels, rs[-1], 3, bias_rule=ZEROS)
    return out


def _int_list(v) -> List[int]:
    """CLI passthrough keywords arrive as strings (experiments_setup/utils.py:40-50 of the reference): accept "2", "1,2", [1, 2]."""
    if isinstance(v, str):
        return [int(x) for x in v.replace("[", "").replace("]", "").split(",") if x.strip()]
    if isinstance(v, int):
        return [v]
    return [int(x) for x in v]


class _Token:
    """lives as long as the autograd node of one forward call (weakly referenced by the graph entry that served the call)"""
    __slots__ = ("__weakref__",)


# Backward phases of a pass that goes through the module's autograd node (the engine's default is one): with its own graphs
# the module runs the leaf work of a phase on a side stream beside the next phase's data-gradient chain, as the pipelined
# driver does.
_MODULE_PHASES = 4

# Graphs of entries that died, parked until a call that is known not to run inside a stream capture destroys them: an
# entry may be released by the garbage collector at any allocation, also in the middle of someone else's capture, where
# destroying a graph is a fatal error of the runtime.
_RETIRED_GRAPHS: list = []


class _GraphEntry:
    """The forward pass and the backward pass of ONE (batch shape, mode, parameter storage) as two HIP graphs that share a
    memory pool, replayed by the module's ordinary autograd path.

    A caller that drives the module the way the reference trainer does (model(x, t) ... loss.backward(), no graph driver)
    pays ~150 kernel launches per step from Python, which is several times the device time of the step.  After one eager
    pass of a shape (it sizes the engine's workspaces and sets pinned staging buffers aside) the module captures the pass
    and from then on a call costs two small copies and one graph launch.  Inputs and the output gradient are copied into
    static buffers; the prediction is returned as a copy; parameter gradients are returned as views of a fresh copy of the
    engine's staging buffer, so autograd may keep or accumulate them.  Everything that changes between calls lives in
    device memory (dropout seeds advance by kernels, the kernel-layout weight copies are refreshed inside the graph)."""

    def __init__(self, module: "BaseUNet", x: Tensor, time: Tensor, record: bool, flat_params):
        # (no reference to the module: the entry must die with it by reference count.  A cycle would leave the graphs to the
        # garbage collector, which may run inside someone else's stream capture, and destroying a graph there is fatal)
        self.record = record
        self.eng = eng = module._get_engine()
        self.names = module._param_names
        self.params = dict(zip(self.names, flat_params))
        self.x = x.clone()
        self.t = time.clone()
        self.pending = False
        self.live = None
        self.g_b = self.dout = self.last_flat = None
        self.plist = list(flat_params)
        self.ptrs = tuple(q.data_ptr() for q in flat_params)          # the graphs bake in every one of these addresses
        # each parameter's gradient-accumulation node: the autograd engine can be asked, during a backward pass, whether it will
        # run them (it will not under torch.autograd.grad or backward(inputs=...))
        with torch.enable_grad():                       # (grad mode is off inside an autograd Function's forward)
            self.acc_nodes = [q.view_as(q).grad_fn.next_functions[0][0] for q in flat_params if q.requires_grad] if record else []
        self.all_need_grad = all(q.requires_grad for q in flat_params)
        repack, eng.always_repack = eng.always_repack, True           # the weight re-layout becomes a node of the graph
        phases, eng.phases = eng.phases, _MODULE_PHASES
        try:
            self.g_f = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_f, capture_error_mode="thread_local"):
                self.out, self.tape = eng.forward(self.params, self.x, self.t, train=module.training, record=record)
        finally:
            eng.always_repack, eng.phases = repack, phases
        # engine-owned buffers the graphs address: they stay alive with the entry even if the engine replaces them later
        self.keep = [eng._zero_arena, eng.grads.flat if eng.grads is not None else None, eng.grads.flat_t if eng.grads is not None else None]

    def __del__(self):
        if _RETIRED_GRAPHS is not None:                 # (None while the interpreter shuts down)
            _RETIRED_GRAPHS.append((getattr(self, "g_f", None), getattr(self, "g_b", None)))

    def usable(self) -> bool:
        """not while an earlier call's autograd graph may still run its backward on the static buffers"""
        return not (self.pending and self.live is not None and self.live() is not None)

    def forward(self, x: Tensor, time: Tensor):
        self.x.copy_(x)
        self.t.copy_(time)
        self.g_f.replay()
        tok = None
        if self.record:
            tok = _Token()
            self.live, self.pending = weakref.ref(tok), True
        return self.out.clone(), tok

    def backward(self, dout: Tensor):
        eng = self.eng
        if self.g_b is None:
            # The backward pass as the engine's phases: per phase a graph with the data-gradient chain (replayed on the
            # caller's stream) and a graph with the phase's leaf work - weight gradients, partial-sum folds, time-embedding
            # MLPs, finally the gradients' re-layout - replayed on a side stream beside the next phase's chain.  Every graph
            # has a memory pool of its own, so nothing a leaf graph still reads is handed to a later chain's allocations.
            from ...runtime.graphs import _pool_stream
            self.dout = dout.contiguous().clone()
            self.side = _pool_stream(self.dout.device, 0)
            nph = len(self.tape[3]) + 1
            self.events = [torch.cuda.Event() for _ in range(nph + 1)]
            chains, leaves = [], []
            for ph in range(nph):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    eng.backward_data(self.tape, ph, self.dout if ph == 0 else None)
                chains.append(g)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self.side, capture_error_mode="thread_local"):
                    eng.backward_flush(self.tape, ph)
                    if ph == nph - 1:
                        eng.grads.unpack_staging()
                leaves.append(g)
            self.g_b = (chains, leaves)
            self.keep += [eng._zero_arena, eng.grads.flat, eng.grads.flat_t]
            lay = eng.grads.tlayout
            self.sizes, self.shapes = [], []
            cursor = 0
            for k in self.names:                        # parameters in registration order, each padded to 4 elements
                o, shp = lay[k]
                assert o == cursor, "gradient staging layout is not the parameters in registration order"
                n = 1
                for d in shp:
                    n *= d
                self.sizes.append((n + 3) // 4 * 4)
                self.shapes.append((n, tuple(shp)))
                cursor += self.sizes[-1]
            self.total = cursor
        else:
            self.dout.copy_(dout)
        main = torch.cuda.current_stream()
        for ph, (chain, leaf) in enumerate(zip(*self.g_b)):
            chain.replay()
            self.events[ph].record(main)
            self.side.wait_event(self.events[ph])
            with torch.cuda.stream(self.side):
                leaf.replay()
        self.events[-1].record(self.side)
        main.wait_event(self.events[-1])
        self.pending = False
        fresh = eng.grads.flat_t[:self.total]
        prev, plist = self.last_flat, self.plist
        if prev is not None and self._engine_accumulates_all() and self._accumulating_into(prev, plist):
            # a later micro-batch of a gradient-accumulation window: every .grad is still the view of the buffer the previous
            # call returned, and this backward pass ends in the accumulation step of every parameter (a plain
            # loss.backward(); under torch.autograd.grad or backward(inputs=subset) the engine hands the gradients to the
            # caller instead, and they must be this call's alone).  autograd would now add the 350 new tensors one launch
            # each; instead the sum is formed here by ONE launch and handed over as this call's gradients, with .grad cleared
            # so that autograd adopts the tensors as they are (hooks on the accumulation step, DistributedDataParallel's
            # among them, run as ever and find the same .grad they would have found)
            flat = torch.add(prev, fresh)
            for q in plist:
                q.grad = None
        else:
            flat = fresh.clone()
        self.last_flat = flat
        return tuple(piece.view(shp) if piece.numel() == n else piece[:n].view(shp)
                     for piece, (n, shp) in zip(flat.split_with_sizes(self.sizes), self.shapes))

    def _engine_accumulates_all(self) -> bool:
        """True when the running backward pass will execute the gradient-accumulation node of EVERY parameter"""
        if not self.all_need_grad or not self.acc_nodes:
            return False
        try:
            will = torch._C._will_engine_execute_node
            return all(will(node) for node in self.acc_nodes)
        except Exception:                               # (the engine raises for a leaf whose gradient it captures for the caller)
            return False

    def _accumulating_into(self, prev: Tensor, plist) -> bool:
        base, off = prev.data_ptr(), 0
        for q, size, (n, shp) in zip(plist, self.sizes, self.shapes):
            g = q.grad
            if g is None or g.data_ptr() != base + 4 * off or g.dtype != torch.float32 or tuple(g.shape) != shp or q._backward_hooks:
                return False
            off += size
        return True


class _UNetFn(torch.autograd.Function):
    """One autograd node for the whole denoiser: forward runs the fused program, backward replays its tape."""

    @staticmethod
    def forward(ctx, module: "BaseUNet", record: bool, x: Tensor, time: Tensor, *flat_params: Tensor):
        eng = module._get_engine()
        ctx.set_materialize_grads(False)
        ctx.module, ctx.entry, ctx.key = module, None, None
        entry, key = module._graph_entry(x, time, record, flat_params)
        if entry is not None:
            out, ctx.token = entry.forward(x, time)
            ctx.entry, ctx.tape = entry, (entry if record else None)
            return out
        params = dict(zip(module._param_names, flat_params))
        phases = eng.phases
        if module.use_graphs:             # every pass of the module takes the backward phases its captured passes take: the first
            eng.phases = _MODULE_PHASES   # pass of a (shape, phases) sizes the engine's workspaces, which no capture may do
        try:
            out, tape = eng.forward(params, x, time, train=module.training, record=record, taps=module._taps)
        finally:
            eng.phases = phases
        ctx.tape, ctx.key = tape, key
        if key is not None and not record:
            module._graph_seen[key] = module._graph_seen.get(key, 0) + 1
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        module = ctx.module
        if ctx.tape is None or dout is None:
            return (None, None, None, None) + (None,) * len(module._param_names)
        if ctx.entry is not None:
            grads = ctx.entry.backward(dout)
            ctx.tape = None
            return (None, None, None, None) + grads
        eng = module._get_engine()
        params = dict(zip(module._param_names, module._parameter_list()))
        grads = eng.backward(ctx.tape, dout, params)
        ctx.tape = None
        if ctx.key is not None:                          # one complete eager pass of this shape: the next one is captured
            module._graph_seen[ctx.key] = module._graph_seen.get(ctx.key, 0) + 1
        return (None, None, None, None) + tuple(grads[k] for k in module._param_names)


class BaseUNet(nn.Module):
    """Drop-in for the reference BaseUNet: `BaseUNet(**kwargs)(input (N,C,H,W) float, time (N,) long) -> (N,C,H,W)`.

    Keyword arguments are the ones the reference forwards (DDPM.yaml + CLI extras, unknown ones are accepted and
    ignored, dm_utils.py:39-58 there).  `T` is NOT forwarded by the reference's DDPM wrapper, so the time
    normaliser defaults to 100 exactly as it does there.

    Extension (no counterpart in the reference, whose `is_conditional` branches are commented out,
    constituent_blocks.py:92-101,194-203): `attention_stages=[i, ...]` inserts a self-attention ConditionalModule
    (condition = the layer's own input, kdim = vdim = C, `attention_depth` layers, default 1) at `layers[1]` of encoder
    stage i's DownBlock and of the decoder UpBlock working at the same resolution - where that commented code would."""

    def __init__(self, **kwargs):
        super().__init__()
        act_name = kwargs.get("activation", "gelu")
        activation_code(act_name)
        self.spec = UNetSpec(
            img_channels=kwargs["img_channels"], stages_channels=kwargs["stages_channels"], latent_dim=kwargs["latent_dim"],
            expand_factor=kwargs.get("expand_factor", 4), drop_p=float(kwargs.get("drop_p", 0.1)),
            t_embed_dim=kwargs.get("t_embed_dim", 32), activation=act_name, base_groups_norm=kwargs["base_groups_norm"],
            time_norm=float(kwargs.get("T", 100)), attention_stages=_int_list(kwargs.get("attention_stages", ())),
            attention_depth=int(kwargs.get("attention_depth", 1)))
        self.is_conditional = bool(kwargs.get("is_conditional", False))
        device, dtype = kwargs.get("device", None), kwargs.get("dtype", None)
        if dtype not in (None, torch.float32):
            raise ValueError("parameters are kept in fp32 (the kernels compute in bf16 with fp32 accumulation)")
        init_w = get_initializer(kwargs.get("initializer", "normal"))
        init_dec_embed = get_initializer("he_uniform")
        self.encoder, self.decoder = _Slot(), _Slot()
        names = []
        for name, shape, rule in parameter_table(self.spec):
            t = torch.empty(shape, device=device, dtype=torch.float32)
            if callable(rule):
                rule(t)
            elif rule == W_INIT:
                init_w(t)
            elif rule == W_INIT_EMBED_DEC:
                init_dec_embed(t)
            elif rule == ONES:
                nn.init.ones_(t)
            elif rule == ZEROS:
                nn.init.zeros_(t)
            else:                                   # nn.Linear's default bias init: U(+-1/sqrt(fan_in))
                bound = 1.0 / math.sqrt(self.spec.t_dim)
                nn.init.uniform_(t, -bound, bound)
            top, _, rest = name.partition(".")
            getattr(self, top).put(rest, nn.Parameter(t))
            names.append(name)
        self._param_names: List[str] = names
        self._engine = None
        self._taps = None
        self._param_slots = None
        import os
        # the autograd path replays captured HIP graphs from the second call of a shape on (see _GraphEntry); set to False
        # (or DDPM_MODULE_GRAPHS=0) to launch every kernel from Python on every call
        self.use_graphs = os.environ.get("DDPM_MODULE_GRAPHS", "1") != "0"
        self._graphs: Dict[tuple, _GraphEntry] = {}
        self._graph_seen: Dict[tuple, int] = {}

    # the engine holds device caches only; it is rebuilt on demand (after deepcopy / .to() / unpickling)
    def _get_engine(self) -> UNetEngine:
        if self._engine is None:
            self._engine = UNetEngine(self.spec)
        return self._engine

    def _graph_entry(self, x: Tensor, time: Tensor, record: bool, flat_params):
        """(captured entry to serve this call with, or None for the eager path; key under which an eager pass is counted)"""
        if (not self.use_graphs or self._taps is not None or not flat_params or not x.is_cuda or x.dim() != 4
                or torch.cuda.is_current_stream_capturing()):
            return None, None
        if _RETIRED_GRAPHS:               # (not capturing here: a safe place to destroy the graphs of dead entries)
            _RETIRED_GRAPHS.clear()
        eng = self._get_engine()
        if eng.mask_ahead:                # a pipelined driver draws this engine's dropout bits one step ahead
            return None, None
        key = (tuple(x.shape), x.device.index, self.training, record, flat_params[0].data_ptr(), flat_params[-1].data_ptr(), eng.phases)
        e = self._graphs.get(key)
        if e is not None and e.ptrs != tuple(q.data_ptr() for q in flat_params):
            # a parameter in the middle was re-pointed (p.data = ..., a partial weight swap): the captured graphs address the
            # old storage.  Drop the entry; the shape is captured again on this call.
            del self._graphs[key]
            e = None
        if e is None:
            for k in [k for k, v in self._graphs.items() if k[4:6] != key[4:6]]:
                del self._graphs[k]                     # entries of a parameter storage that is gone (each holds its activations)
            if self._graph_seen.get(key, 0) < 1:
                return None, key
            if not x.is_contiguous() or time.dtype != torch.long or time.dim() != 1:
                return None, None
            while len(self._graphs) >= 3:              # a few shapes / modes at most stay captured (each holds its activations)
                self._graphs.pop(next(iter(self._graphs)))
            e = self._graphs[key] = _GraphEntry(self, x, time, record, flat_params)
        return (e if e.usable() else None), None

    def __deepcopy__(self, memo):
        eng, self._engine = self._engine, None
        graphs, seen, self._graphs, self._graph_seen = self._graphs, self._graph_seen, {}, {}
        slots, self._param_slots = self._param_slots, None
        try:
            cls = self.__class__
            new = cls.__new__(cls)
            memo[id(self)] = new
            import copy
            for k, v in self.__dict__.items():
                setattr(new, k, copy.deepcopy(v, memo))
        finally:
            self._engine = eng
            self._graphs, self._graph_seen = graphs, seen
            self._param_slots = slots
        return new

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_engine"] = None
        st["_graphs"], st["_graph_seen"] = {}, {}
        st["_param_slots"] = None
        return st

    def _parameter_list(self) -> List[Tensor]:
        """The parameters in registration order, read from the containers' own tables: nn.Module.parameters() walks the module
        tree with its bookkeeping on every call (~0.9 ms for the 350 tensors), which a training loop pays per micro-batch."""
        if self._param_slots is None:
            slots = []
            for name in self._param_names:
                mod = self
                *path, leaf = name.split(".")
                for part in path:
                    mod = mod._modules[part]
                slots.append((mod._parameters, leaf))
            self._param_slots = slots
        return [table[leaf] for table, leaf in self._param_slots]

    def forward(self, input: Tensor, time: Tensor, *args, **kwargs) -> Tensor:
        if not input.is_cuda:
            raise RuntimeError("BaseUNet runs on the HIP kernels only: move the module and its inputs to an MI355X (cuda) device")
        params = self._parameter_list()
        # grad mode is off inside Function.forward, so decide here whether the tape must be recorded
        record = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return _UNetFn.apply(self, record, input.float(), time, *params)
