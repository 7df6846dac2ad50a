"""Shared plumbing of the block-level modules (SeparableInvertResidual, DownBlock, UpBlock, EmbedBlock): fp32
nn.Parameters under the reference's names, one autograd node per block call, arithmetic on the HIP kernels through
the same fused units the whole-network engine uses (runtime/engine.py)."""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
from torch import Tensor, nn

from ..runtime import hip
from ..runtime.engine import UNetEngine
from ..utils.initializers import get_initializer
from .activations import activation_code, has_scalar

W_INIT, ONES, ZEROS, LINEAR_BIAS_DEFAULT = range(4)
BUF_ZEROS, BUF_ONES, BUF_COUNT = "buffer:zeros", "buffer:ones", "buffer:count"      # BatchNorm2d's registered buffers
BF16 = torch.bfloat16


class BlockSpec:
    """What the engine needs to know about a block: its convolutions (name -> kind), depthwise convolutions, and the
    shared hyper-parameters (constituent_blocks.py:51-61 / conv_block.py:20-33 of the reference)."""

    def __init__(self, convs: Dict[str, int], dwconvs: List[str], expand_factor, drop_p: float, activation: str, base_groups_norm: int):
        self.convs, self.dwconvs = convs, dwconvs
        self.expand = expand_factor
        self.drop_p = float(drop_p)
        self.act = hip.ACT[activation]
        self.act_name = activation
        self.act_scalar = self.act in hip.ACT_WITH_SCALAR
        self.groups = 4 * int(base_groups_norm)


class _Slot(nn.Module):
    """Parameter container addressed by dotted paths ('layers.1.expansion.1.weight')."""

    def put(self, path: str, p) -> None:
        """p: an nn.Parameter, or a plain tensor to be registered as a buffer"""
        head, _, rest = path.partition(".")
        if not rest:
            if isinstance(p, nn.Parameter):
                self.register_parameter(head, p)
            else:
                self.register_buffer(head, p)
            return
        if head not in self._modules:
            self.add_module(head, _Slot())
        self._modules[head].put(rest, p)


def nchw_to_kernel(x: Tensor) -> Tensor:
    """(N,C,H,W) fp32 -> NHWC bf16 on the layout-bridge kernel"""
    x = x.float().contiguous()
    n, c, h, w = x.shape
    out = torch.empty(n, h, w, c, device=x.device, dtype=BF16)
    hip.call("ddpm_nchw_f32_to_nhwc_bf16", hip.ptr(x), hip.ptr(out), n, c, h * w, c, hip.stream())
    return out


def kernel_to_nchw(x: Tensor) -> Tensor:
    n, h, w, c = x.shape
    out = torch.empty(n, c, h, w, device=x.device, dtype=torch.float32)
    hip.call("ddpm_nhwc_bf16_to_nchw_f32", hip.ptr(x), hip.ptr(out), n, c, h * w, c, hip.stream())
    return out


class _BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module: "BlockModule", record: bool, n_in: int, *tensors: Tensor):
        inputs, flat_params = tensors[:n_in], tensors[n_in:]
        eng = module._get_engine()
        params = dict(zip(module._param_names, flat_params))
        params.update(module.named_buffers())          # BatchNorm2d statistics: read (and, in training mode, advanced) by the kernels
        ctx.set_materialize_grads(False)
        kin = [module._to_kernel(t) for t in inputs]
        key = (module.__class__.__name__,) + tuple(tuple(t.shape) for t in kin)
        out, tape = eng.run_program(params, kin, [t.requires_grad for t in inputs], module._program, key, module.training, record,
                                    lambda: module._site_sizes([tuple(t.shape) for t in kin]))
        ctx.tape, ctx.module, ctx.n_in = tape, module, n_in
        return module._from_kernel(out)

    @staticmethod
    def backward(ctx, dout: Tensor):
        module, n_in = ctx.module, ctx.n_in
        nothing = (None, None, None) + (None,) * (n_in + len(module._param_names))
        if ctx.tape is None or dout is None:
            return nothing
        gin, gp = module._get_engine().backward_program(ctx.tape, module._grad_to_kernel(dout))
        ctx.tape = None
        return (None, None, None) + tuple(None if g is None else module._grad_from_kernel(g) for g in gin) + tuple(gp[k] for k in module._param_names)


class BlockModule(nn.Module):
    """Base: subclasses give parameter_table(), a BlockSpec, _program(units, inputs) and _site_sizes(input shapes)."""

    def _build(self, table: Sequence[Tuple[str, Tuple[int, ...], int]], spec: BlockSpec, initializer, device, dtype, linear_fan_in: int = 1):
        if dtype not in (None, torch.float32):
            raise ValueError("parameters are kept in fp32 (the kernels compute in bf16 with fp32 accumulation)")
        init_w = initializer if callable(initializer) else get_initializer(initializer)
        self._spec = spec
        names = []
        for name, shape, rule in table:
            if rule in (BUF_ZEROS, BUF_ONES, BUF_COUNT):
                t = (torch.zeros(shape, device=device, dtype=torch.long) if rule == BUF_COUNT else
                     torch.full(shape, 1.0 if rule == BUF_ONES else 0.0, device=device, dtype=torch.float32))
                head, _, rest = name.partition(".")
                if head not in self._modules:
                    self.add_module(head, _Slot())
                self._modules[head].put(rest, t)
                continue
            t = torch.empty(shape, device=device, dtype=torch.float32)
            if callable(rule):
                rule(t)
            elif rule == W_INIT:
                init_w(t)
            elif rule == ONES:
                nn.init.ones_(t)
            elif rule == ZEROS:
                nn.init.zeros_(t)
            else:
                bound = 1.0 / math.sqrt(linear_fan_in)
                nn.init.uniform_(t, -bound, bound)
            head, _, rest = name.partition(".")
            if head not in self._modules:
                self.add_module(head, _Slot())
            self._modules[head].put(rest, nn.Parameter(t))
            names.append(name)
        self._param_names: List[str] = names
        self._engine = None

    def _get_engine(self) -> UNetEngine:
        if self._engine is None:
            self._engine = UNetEngine(self._spec)
        return self._engine

    def __deepcopy__(self, memo):
        import copy
        eng, self._engine = self._engine, None
        try:
            new = self.__class__.__new__(self.__class__)
            memo[id(self)] = new
            for k, v in self.__dict__.items():
                setattr(new, k, copy.deepcopy(v, memo))
        finally:
            self._engine = eng
        return new

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_engine"] = None
        return st

    # layout bridges (4-D blocks; EmbedBlock overrides)
    _to_kernel = staticmethod(nchw_to_kernel)
    _from_kernel = staticmethod(kernel_to_nchw)
    _grad_to_kernel = staticmethod(nchw_to_kernel)
    _grad_from_kernel = staticmethod(kernel_to_nchw)

    def _run(self, *inputs: Tensor) -> Tensor:
        if not inputs[0].is_cuda:
            raise RuntimeError(f"{self.__class__.__name__} runs on the HIP kernels only: move the module and its inputs to an MI355X (cuda) device")
        params = [p for p in self.parameters()]
        record = torch.is_grad_enabled() and (any(p.requires_grad for p in params) or any(t.requires_grad for t in inputs))
        return _BlockFn.apply(self, record, len(inputs), *inputs, *params)


def conv_entries(name: str, co: int, ci: int, k: int):
    return [(name + ".weight", (co, ci, k, k), W_INIT), (name + ".bias", (co,), ONES)]


def norm_entries(name: str, c: int):
    return [(name + ".weight", (c,), ONES), (name + ".bias", (c,), ZEROS)]


def bn_entries(name: str, c: int) -> list:
    """torch.nn.BatchNorm2d(c): weight = 1, bias = 0 and the buffers running_mean = 0, running_var = 1, num_batches_tracked = 0"""
    return norm_entries(name, c) + [(name + ".running_mean", (c,), BUF_ZEROS), (name + ".running_var", (c,), BUF_ONES),
                                    (name + ".num_batches_tracked", (), BUF_COUNT)]


def act_entries(site: str, activation: str) -> list:
    """the activation module at `site`: swish / hardswish own one 0-dim parameter `beta`, initialised to 1
    (neural_nets/activations.py:40,60 of the reference); the others own nothing"""
    return [(site + ".beta", (), ONES)] if has_scalar(activation) else []


def sir_entries(pre: str, c: int, expand, activation: str = "gelu", out_channels: int = 0) -> list:
    e = int(expand * c)
    co = out_channels or c
    return (conv_entries(pre + "expansion.1", e, c, 1) + norm_entries(pre + "expansion.2", e) + act_entries(pre + "expansion.3", activation) +
            conv_entries(pre + "depthwise.1", e, 1, 3) + norm_entries(pre + "depthwise.2", e) + act_entries(pre + "depthwise.3", activation) +
            conv_entries(pre + "projection.1", co, e, 1) + norm_entries(pre + "projection.2", co))


def check_activation(name: str) -> None:
    activation_code(name)
