"""DownBlock / UpBlock of the lightweight autoencoder behind the reference's constructors
(neural_nets/autoencoders/constituent_blocks.py:20-198 there), executed on the HIP kernels.

    DownBlock: [conv3x3 s2 (no bias) + BatchNorm2d + act] -> SeparableInvertResidual -> [conv3x3 (no bias) + BatchNorm2d + act]   (:56-88)
    UpBlock:   [convT 2x2 s2 (no bias) + BatchNorm2d + act] -> SeparableInvertResidual -> [conv3x3 (no bias) + BatchNorm2d + act] (:150-181)

state_dict keys as there: layers.0.weight, layers.1.{weight,bias,running_mean,running_var,num_batches_tracked},
[layers.2.beta], layers.3.{expansion,depthwise,projection}.*, layers.4.weight, layers.5.*, [layers.6.beta] (the beta entries
exist for swish / hardswish, whose modules own a learnable scalar; hardswish is these blocks' default).

BatchNorm2d runs on the GroupNorm kernels: in training mode its statistics are those of a GroupNorm with one group per
channel on the batch viewed as one image; the running statistics advance with torch's momentum rule and are the fixed
statistics of eval mode."""
from __future__ import annotations

from torch import Tensor

from ...runtime import hip
from .._block_base import BlockModule, BlockSpec, act_entries, bn_entries, check_activation, sir_entries, W_INIT


def _conv_nobias(name: str, co: int, ci: int, k: int, transposed: bool = False):
    return [(name + ".weight", (ci, co, k, k) if transposed else (co, ci, k, k), W_INIT)]


def _sir_convs(pre: str):
    return {pre + "expansion.1": hip.CONV_1X1, pre + "projection.1": hip.CONV_1X1}


def down_entries(pre: str, ci: int, co: int, expand, act: str):
    return (_conv_nobias(pre + "layers.0", co, ci, 3) + bn_entries(pre + "layers.1", co) + act_entries(pre + "layers.2", act) +
            sir_entries(pre + "layers.3.", co, expand, act) +
            _conv_nobias(pre + "layers.4", co, co, 3) + bn_entries(pre + "layers.5", co) + act_entries(pre + "layers.6", act))


def down_convs(pre: str):
    return dict({pre + "layers.0": hip.CONV_3X3_S2, pre + "layers.4": hip.CONV_3X3_S1}, **_sir_convs(pre + "layers.3."))


def down_program(u, pre: str, z, act: int):
    z = u.conv_bn_unit(pre + "layers.0", z, pre + "layers.1", act)
    z = u.sir(pre + "layers.3.", z)
    return u.conv_bn_unit(pre + "layers.4", z, pre + "layers.5", act)


def up_entries(pre: str, ci: int, co: int, expand, act: str):
    return (_conv_nobias(pre + "layers.0", ci, ci, 2, transposed=True) + bn_entries(pre + "layers.1", ci) + act_entries(pre + "layers.2", act) +
            sir_entries(pre + "layers.3.", ci, expand, act) +
            _conv_nobias(pre + "layers.4", co, ci, 3) + bn_entries(pre + "layers.5", co) + act_entries(pre + "layers.6", act))


def up_convs(pre: str):
    return dict({pre + "layers.0": hip.CONVT_2X2_S2, pre + "layers.4": hip.CONV_3X3_S1}, **_sir_convs(pre + "layers.3."))


def up_program(u, pre: str, z, act: int):
    z = u.conv_bn_unit(pre + "layers.0", z, pre + "layers.1", act)
    z = u.sir(pre + "layers.3.", z)
    return u.conv_bn_unit(pre + "layers.4", z, pre + "layers.5", act)


def sir_sites(n: int, h: int, w: int, c: int, expand) -> list:
    """element counts of the three dropout sites of an inverted-residual block on an [n, h, w, c] tensor"""
    e = int(expand * c)
    return [n * h * w * c, n * h * w * e, n * h * w * e]


class DownBlock(BlockModule):
    def __init__(self, in_channels: int, out_channels: int, expand_factor: int = 3, drop_p: float = 0.3, activation: str = "hardswish",
                 initializer="he_uniform", device=None, dtype=None, *args, **kwargs):
        super().__init__()
        check_activation(activation)
        if in_channels % 16 or out_channels % 32:
            raise ValueError(f"channels must be multiples of 16 (in) / 32 (out) for the MFMA kernels, got {in_channels} -> {out_channels}")
        self.expand, self.cout = expand_factor, out_channels
        spec = BlockSpec(down_convs(""), ["layers.3.depthwise.1"], expand_factor, drop_p, activation, 4)
        self._build(down_entries("", in_channels, out_channels, expand_factor, activation), spec, initializer, device, dtype)

    def _program(self, u, inputs):
        return down_program(u, "", inputs[0], self._spec.act)

    def _site_sizes(self, shapes):
        n, h, w, _ = shapes[0]
        return sir_sites(n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, self.cout, self.expand)

    def forward(self, input: Tensor) -> Tensor:
        return self._run(input)


class UpBlock(BlockModule):
    def __init__(self, in_channels: int, out_channels: int, expand_factor: int = 3, drop_p: float = 0.3, activation: str = "hardswish",
                 initializer="he_uniform", device=None, dtype=None):
        super().__init__()
        check_activation(activation)
        if in_channels % 32 or out_channels % 32:
            raise ValueError(f"channels must be multiples of 32 for the MFMA kernels, got {in_channels} -> {out_channels}")
        self.expand, self.cin = expand_factor, in_channels
        spec = BlockSpec(up_convs(""), ["layers.3.depthwise.1"], expand_factor, drop_p, activation, 4)
        self._build(up_entries("", in_channels, out_channels, expand_factor, activation), spec, initializer, device, dtype)

    def _program(self, u, inputs):
        return up_program(u, "", inputs[0], self._spec.act)

    def _site_sizes(self, shapes):
        n, h, w, _ = shapes[0]
        return sir_sites(n, 2 * h, 2 * w, self.cin, self.expand)

    def forward(self, input: Tensor) -> Tensor:
        return self._run(input)
