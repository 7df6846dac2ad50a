"""SeparableInvertResidual behind the reference's constructor (neural_nets/conv_block.py:19-136 there): the
MobileNetV2-style block dropout -> 1x1 expansion -> GroupNorm -> act -> dropout -> depthwise 3x3 -> GroupNorm -> act ->
dropout -> 1x1 projection -> GroupNorm (+ input), executed by one fused HIP kernel per direction where the shape allows
and by the per-unit kernels otherwise.  state_dict keys: expansion.{1,2}.*, depthwise.{1,2}.*, projection.{1,2}.*, plus
expansion.3.beta and depthwise.3.beta when the activation is swish or hardswish (the reference block's default is swish)."""
from __future__ import annotations

from ..runtime import hip
from ._block_base import BlockModule, BlockSpec, check_activation, sir_entries


class SeparableInvertResidual(BlockModule):
    def __init__(self, in_channels: int, out_channels: int, expand_factor: int, stride: int = 1, drop_p: float = 0.3,
                 base_groups_norm: int = 4, activation: str = "swish", initializer="he_uniform", device=None, dtype=None, *args, **kwargs):
        super().__init__()
        check_activation(activation)
        if stride not in (1, 2):
            raise ValueError(f"the depthwise convolution's stride is 1 or 2, got {stride}")
        if in_channels % 32 or out_channels % 32:
            raise ValueError(f"channel counts must be multiples of 32 for the MFMA kernels, got {in_channels} -> {out_channels}")
        # conv_block.py:51-54 of the reference: the block adds its input only when it keeps the shape.  The residual form is the
        # one the reference's UNet builds (constituent_blocks.py:77-80,186-192) and the one with a single-launch kernel; the
        # others run on the per-unit kernels.
        self.is_residual = in_channels == out_channels and stride == 1
        self.in_channels, self.out_channels, self.stride, self.expand_factor = in_channels, out_channels, int(stride), expand_factor
        spec = BlockSpec({"expansion.1": hip.CONV_1X1, "projection.1": hip.CONV_1X1}, ["depthwise.1"], expand_factor, drop_p, activation,
                         base_groups_norm)
        self._build(sir_entries("", in_channels, expand_factor, activation, out_channels), spec, initializer, device, dtype)

    def _program(self, units, inputs):
        return units.sir("", inputs[0], residual=self.is_residual, stride=self.stride)

    def _site_sizes(self, shapes):
        n, h, w, c = shapes[0]
        e = int(self.expand_factor * c)
        ho, wo = (h + self.stride - 1) // self.stride, (w + self.stride - 1) // self.stride
        return [n * h * w * c, n * h * w * e, n * ho * wo * e]

    def forward(self, input):
        return self._run(input)
