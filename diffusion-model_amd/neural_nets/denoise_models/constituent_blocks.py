"""DownBlock, UpBlock, EmbedBlock and SinusoidalPositionEmbedBlock behind the reference's constructors
(neural_nets/denoise_models/constituent_blocks.py:49-328 there), executed on the HIP kernels.

    DownBlock:  [conv3x3 s2 + GroupNorm + act] -> SIR -> SIR -> [conv3x3 + GroupNorm + act]             (:63-90,135-147)
    UpBlock:    cat(input, encoder_output) -> [conv3x3 + GN + act] -> SIR -> [convT 2x2 s2 + GN + act] -> SIR -> SIR  (:165-193,227-243)
    EmbedBlock: Linear(D -> 2C) -> act -> Linear(2C -> C) -> (N, C, 1, 1)                                  (:308-328)

    ConditionalModule: depth x { MultiheadAttention(8 heads) -> GroupNorm(1,C)(. + in) -> Linear -> act -> GroupNorm(1,C)(. + prev) }  (:331-437)

The reference's Down/Up blocks never insert the ConditionalModule (the insertion is commented out there, :92-101,194-203);
BaseUNet's `attention_stages` keyword places it where that commented code would, as self-attention."""
from __future__ import annotations

import torch
from torch import Tensor, nn

import math

from ...runtime import hip
from ...runtime.engine import attention_convs
from .._block_base import (ZEROS, BF16, LINEAR_BIAS_DEFAULT, ONES, W_INIT, BlockModule, BlockSpec, act_entries, check_activation, conv_entries,
                           norm_entries, sir_entries)


def _common(kwargs):
    return (kwargs.get("expand_factor", 4), kwargs.get("drop_p", 0.1), kwargs.get("activation", "gelu"), kwargs.get("base_groups_norm", 4),
            kwargs.get("initializer", "normal"), kwargs.get("device", None), kwargs.get("dtype", None))


class DownBlock(BlockModule):
    def __init__(self, in_channels: int, out_channels: int, is_conditional: bool = False, **kwargs):
        super().__init__()
        expand, drop_p, act, bgn, init, device, dtype = _common(kwargs)
        check_activation(act)
        if in_channels % 16 or out_channels % 32:
            raise ValueError(f"channels must be multiples of 16 (in) / 32 (out) for the MFMA kernels, got {in_channels} -> {out_channels}")
        self.expand = expand
        co = out_channels
        table = (conv_entries("layers.0.0", co, in_channels, 3) + norm_entries("layers.0.1", co) + act_entries("layers.0.2", act) +
                 sir_entries("layers.1.", co, expand, act) + sir_entries("layers.2.", co, expand, act) +
                 conv_entries("layers.3.0", co, co, 3) + norm_entries("layers.3.1", co) + act_entries("layers.3.2", act))
        convs = {"layers.0.0": hip.CONV_3X3_S2, "layers.1.expansion.1": hip.CONV_1X1, "layers.1.projection.1": hip.CONV_1X1,
                 "layers.2.expansion.1": hip.CONV_1X1, "layers.2.projection.1": hip.CONV_1X1, "layers.3.0": hip.CONV_3X3_S1}
        self._build(table, BlockSpec(convs, ["layers.1.depthwise.1", "layers.2.depthwise.1"], expand, drop_p, act, bgn), init, device, dtype)

    def _program(self, u, inputs):
        sp = self._spec
        z = u.conv_unit("layers.0.0", inputs[0], norm="layers.0.1", groups=sp.groups, act=sp.act)
        z = u.sir("layers.1.", z)
        z = u.sir("layers.2.", z)
        return u.conv_unit("layers.3.0", z, norm="layers.3.1", groups=sp.groups, act=sp.act)

    def _site_sizes(self, shapes):
        n, h, w, _ = shapes[0]
        h2, w2 = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        c = self._modules["layers"]._modules["3"]._modules["0"].bias.shape[0]
        e = int(self.expand * c)
        return [n * h2 * w2 * c, n * h2 * w2 * e, n * h2 * w2 * e] * 2

    def forward(self, input: Tensor, *args, **kwargs) -> Tensor:
        return self._run(input)


class UpBlock(BlockModule):
    def __init__(self, in_channels: int, out_channels: int, is_conditional: bool = False, **kwargs):
        super().__init__()
        expand, drop_p, act, bgn, init, device, dtype = _common(kwargs)
        check_activation(act)
        if in_channels % 32 or out_channels % 32:
            raise ValueError(f"channels must be multiples of 32 for the MFMA kernels, got {in_channels} -> {out_channels}")
        self.expand, self.cin, self.cout = expand, in_channels, out_channels
        ci, co = in_channels, out_channels
        table = (conv_entries("layers.0.0", ci, 2 * ci, 3) + norm_entries("layers.0.1", ci) + act_entries("layers.0.2", act) +
                 sir_entries("layers.1.", ci, expand, act) +
                 [("layers.2.0.weight", (ci, co, 2, 2), W_INIT), ("layers.2.0.bias", (co,), ONES)] + norm_entries("layers.2.1", co) +
                 act_entries("layers.2.2", act) + sir_entries("layers.3.", co, expand, act) + sir_entries("layers.4.", co, expand, act))
        convs = {"layers.0.0": hip.CONV_3X3_S1, "layers.2.0": hip.CONVT_2X2_S2}
        for i in (1, 3, 4):
            convs[f"layers.{i}.expansion.1"] = hip.CONV_1X1
            convs[f"layers.{i}.projection.1"] = hip.CONV_1X1
        self._build(table, BlockSpec(convs, [f"layers.{i}.depthwise.1" for i in (1, 3, 4)], expand, drop_p, act, bgn), init, device, dtype)

    def _program(self, u, inputs):
        sp = self._spec
        z = u.conv_unit("layers.0.0", inputs[0], x1=inputs[1], norm="layers.0.1", groups=sp.groups, act=sp.act)
        z = u.sir("layers.1.", z)
        z = u.conv_unit("layers.2.0", z, norm="layers.2.1", groups=sp.groups, act=sp.act)
        z = u.sir("layers.3.", z)
        return u.sir("layers.4.", z)

    def _site_sizes(self, shapes):
        n, h, w, _ = shapes[0]
        ci, co, ex = self.cin, self.cout, self.expand
        lo, hi = n * h * w, n * 4 * h * w
        return [lo * ci, lo * int(ex * ci), lo * int(ex * ci)] + [hi * co, hi * int(ex * co), hi * int(ex * co)] * 2

    def forward(self, input: Tensor, encoder_output: Tensor, *args, **kwargs) -> Tensor:
        return self._run(input, encoder_output)


def _linear_default(fan_in: int):
    """nn.Linear's default initialisation of weight and bias: U(-1/sqrt(fan_in), 1/sqrt(fan_in))"""
    bound = 1.0 / math.sqrt(fan_in)
    return lambda t: nn.init.uniform_(t, -bound, bound)


def conditional_entries(pre: str, c: int, depth: int, kdim: int, activation: str = "gelu"):
    """Parameters of a ConditionalModule in the reference's registration order (constituent_blocks.py:357-399; inside
    nn.MultiheadAttention: in_proj_weight or q/k/v_proj_weight, in_proj_bias, out_proj) with torch's default initialisers
    (xavier_uniform for the input projections, zero attention biases, nn.Linear defaults elsewhere)."""
    out = []
    for i in range(depth):
        m = f"{pre}mhas.{i}."
        if kdim == c:
            out.append((m + "in_proj_weight", (3 * c, c), nn.init.xavier_uniform_))
        else:
            out += [(m + "q_proj_weight", (c, c), nn.init.xavier_uniform_), (m + "k_proj_weight", (c, kdim), nn.init.xavier_uniform_),
                    (m + "v_proj_weight", (c, kdim), nn.init.xavier_uniform_)]
        out += [(m + "in_proj_bias", (3 * c,), ZEROS), (m + "out_proj.weight", (c, c), _linear_default(c)), (m + "out_proj.bias", (c,), ZEROS)]
    for i in range(depth):
        out += norm_entries(f"{pre}mha_norms.{i}", c)
    for i in range(depth):
        out += [(f"{pre}linears.{i}.weight", (c, c), _linear_default(c)), (f"{pre}linears.{i}.bias", (c,), _linear_default(c))]
    for i in range(depth):
        out += norm_entries(f"{pre}linear_norms.{i}", c)
    for i in range(depth):                              # self.activations is registered last (:398-399)
        out += act_entries(f"{pre}activations.{i}", activation)
    return out


class ConditionalModule(BlockModule):
    """Drop-in for the reference ConditionalModule (constituent_blocks.py:331-437): forward(querry [N, S, C], condition
    [N, Sk, kdim]) -> [N, S, C].  Passing the same tensor as both arguments is self-attention (one packed projection GEMM).
    The reference passes `condition` as both key and value, so kdim must equal vdim."""

    def __init__(self, in_channels: int, activation: str, depth, drop_p: float, kdim, vdim, device=None, dtype=None):
        super().__init__()
        check_activation(activation)
        c, kdim, vdim = int(in_channels), int(kdim), int(vdim)
        if kdim != vdim:
            raise ValueError(f"condition is used as key and value, so kdim ({kdim}) must equal vdim ({vdim})")
        if c % 32 or kdim % 16 or c // 8 not in (4, 8, 16, 24, 32, 40, 48, 56, 64):
            raise ValueError(f"in_channels must be a multiple of 32 with a head width (in_channels / 8) the attention kernel supports, "
                             f"kdim a multiple of 16; got {c}, {kdim}")
        self.in_channels, self.kdim, self.depth = c, kdim, int(depth)
        convs, aliases = {}, {}
        attention_convs(convs, aliases, "", c, kdim, self.depth)
        spec = BlockSpec(convs, [], 2, drop_p, activation, 4)
        spec.aliases = aliases
        self._build(conditional_entries("", c, self.depth, kdim, activation), spec, "normal", device, dtype)

    def _program(self, u, inputs):
        cond = inputs[1] if len(inputs) > 1 else None
        return u.attn_module("", inputs[0], cond, self.depth)

    def _site_sizes(self, shapes):
        return []

    @staticmethod
    def _to_kernel(x: Tensor) -> Tensor:
        return x.to(BF16).contiguous().unsqueeze(2)

    @staticmethod
    def _from_kernel(o: Tensor) -> Tensor:
        return o.squeeze(2).float()

    _grad_to_kernel = _to_kernel
    _grad_from_kernel = _from_kernel

    def forward(self, querry: Tensor, condition: Tensor, **kwargs) -> Tensor:
        if kwargs:
            raise ValueError(f"attention keyword arguments are not supported by the fused kernel: {sorted(kwargs)}")
        if condition is querry:
            return self._run(querry)
        return self._run(querry, condition)


class SinusoidalPositionEmbedBlock(nn.Module):
    """time (N,) -> (N, dim) = [sin(t f_i), cos(t f_i)], f_i = 10000^(-i / (dim/2 - 1))  (constituent_blocks.py:246-279)."""

    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim

    def forward(self, time: Tensor) -> Tensor:
        if not time.is_cuda:
            raise RuntimeError("SinusoidalPositionEmbedBlock runs on the HIP kernels only")
        # the kernel takes integer timesteps and a normaliser; callers of the reference pass t / normaliser as floats,
        # so evaluate in float here with the same formula (fp32) - this block is one tiny elementwise op
        half = self.dim // 2
        import math
        freq = torch.exp(torch.arange(half, device=time.device, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
        arg = time.float().view(-1, 1) * freq[None, :]
        return torch.cat((arg.sin(), arg.cos()), dim=-1)


class EmbedBlock(BlockModule):
    def __init__(self, input_dim: int, embed_dim: int, activation: str = "hardswish", initializer="he_uniform", device=None, dtype=None):
        super().__init__()
        check_activation(activation)
        if input_dim % 16 or embed_dim % 32:
            raise ValueError(f"input_dim must be a multiple of 16 and embed_dim of 32 for the MFMA kernels, got {input_dim}, {embed_dim}")
        self.input_dim, self.embed_dim = input_dim, embed_dim
        table = ([("layers.0.weight", (2 * embed_dim, input_dim), W_INIT), ("layers.0.bias", (2 * embed_dim,), LINEAR_BIAS_DEFAULT)] +
                 act_entries("layers.1", activation) +
                 [("layers.2.weight", (embed_dim, 2 * embed_dim), W_INIT), ("layers.2.bias", (embed_dim,), ONES)])
        spec = BlockSpec({"layers.0": hip.CONV_1X1, "layers.2": hip.CONV_1X1}, [], 2, 0.0, activation, 4)
        self._build(table, spec, initializer, device, dtype, linear_fan_in=input_dim)
        # second Linear's default weight init is replaced by the initializer, its bias by ones (constituent_blocks.py:321-324)

    def _program(self, u, inputs):
        return u.embed("", inputs[0])

    def _site_sizes(self, shapes):
        return []

    @staticmethod
    def _to_kernel(x: Tensor) -> Tensor:
        return x.to(BF16).contiguous().view(x.shape[0], 1, 1, x.shape[1])

    @staticmethod
    def _from_kernel(o: Tensor) -> Tensor:
        return o.view(o.shape[0], o.shape[1], 1, 1)

    @staticmethod
    def _grad_to_kernel(g: Tensor) -> Tensor:
        return g.float().reshape(g.shape[0], -1).contiguous()

    @staticmethod
    def _grad_from_kernel(g: Tensor) -> Tensor:
        return g.float().view(g.shape[0], -1)

    def forward(self, input: Tensor) -> Tensor:
        return self._run(input)
