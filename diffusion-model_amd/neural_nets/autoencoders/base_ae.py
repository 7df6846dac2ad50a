"""Encoder / Decoder of the lightweight autoencoder behind the reference's constructors
(neural_nets/autoencoders/base_ae.py:27-259 there), executed on the HIP kernels.

    Encoder: [conv3x3 (img -> down[0]/2) + act] -> DownBlock(down[0]/2 -> down[0]) -> DownBlock(down[i-1] -> down[i]) ...       (:81-104)
    Decoder: convT 2x2 s2 (latent -> latent) -> act -> conv1x1 (no bias) -> BatchNorm2d -> act -> UpBlock(latent -> up[0]) ->
             UpBlock(up[i-1] -> up[i]) ... -> [conv3x3 + act + conv3x3 (up[-1] -> img)]                                       (:173-228)

The Encoder constructor accepts latent_dim like the reference and, like it, does not use it (its output has down[-1]
channels).  Inputs and outputs are NCHW fp32; parameters keep the reference's state_dict keys and registration order."""
from __future__ import annotations

from typing import Sequence

import torch
from torch import Tensor

from ...runtime import hip
from .._block_base import BF16, ONES, ZEROS, BlockModule, BlockSpec, act_entries, bn_entries, check_activation, conv_entries, W_INIT
from .constituent_blocks import down_convs, down_entries, down_program, sir_sites, up_convs, up_entries, up_program


class Encoder(BlockModule):
    def __init__(self, img_channels: int, down_channels: Sequence[int], latent_dim: int, expand_factor: int, drop_p: float = 0.3,
                 activation: str = "hardswish", device=None, dtype=None, initializer="he_uniform", *args, **kwargs):
        super().__init__()
        check_activation(activation)
        down = [int(c) for c in down_channels]
        if img_channels > 16 or (down[0] // 2) % 16 or any(c % 32 for c in down):
            raise ValueError(f"img_channels <= 16, down_channels[0] / 2 a multiple of 16 and every stage a multiple of 32 are required, "
                             f"got {img_channels}, {down}")
        self.img_channels, self.down_channels, self.latent_dim, self.last_dim, self.expand = img_channels, down, latent_dim, down[-1], expand_factor
        table = conv_entries("layers.0.0", down[0] // 2, img_channels, 3) + act_entries("layers.0.1", activation)
        convs, dws = {"layers.0.0": hip.CONV_3X3_S1}, []
        for i, co in enumerate(down):
            pre = f"layers.{i + 1}."
            table += down_entries(pre, down[0] // 2 if i == 0 else down[i - 1], co, expand_factor, activation)
            convs.update(down_convs(pre))
            dws.append(pre + "layers.3.depthwise.1")
        self._build(table, BlockSpec(convs, dws, expand_factor, drop_p, activation, 4), initializer, device, dtype)

    def _program(self, u, inputs):
        act = self._spec.act
        z = u.conv_unit("layers.0.0", inputs[0], act=act, c0_used=self.img_channels if self.img_channels <= 4 else 0)
        for i in range(len(self.down_channels)):
            z = down_program(u, f"layers.{i + 1}.", z, act)
        return z

    def _site_sizes(self, shapes):
        n, h, w, _ = shapes[0]
        out = []
        for c in self.down_channels:
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            out += sir_sites(n, h, w, c, self.expand)
        return out

    @staticmethod
    def _to_kernel(x: Tensor) -> Tensor:
        """(N,C,H,W) fp32 -> NHWC bf16 with the channels zero-padded to 16 (the stem convolution's input width)"""
        x = x.float().contiguous()
        n, c, h, w = x.shape
        out = torch.empty(n, h, w, 16, device=x.device, dtype=BF16)
        hip.call("ddpm_nchw_f32_to_nhwc_bf16", hip.ptr(x), hip.ptr(out), n, c, h * w, 16, hip.stream())
        return out

    def _grad_from_kernel(self, g: Tensor) -> Tensor:
        return g[..., :self.img_channels].permute(0, 3, 1, 2).float().contiguous()

    def forward(self, input: Tensor) -> Tensor:
        return self._run(input)


class Decoder(BlockModule):
    def __init__(self, img_channels: int, latent_dim: int, up_channels: Sequence[int], expand_factor: int, drop_p: float = 0.3,
                 activation: str = "hardswish", initializer="he_uniform", device=None, dtype=None, *args, **kwargs):
        super().__init__()
        check_activation(activation)
        up = [int(c) for c in up_channels]
        if img_channels > 16 or latent_dim % 32 or any(c % 32 for c in up):
            raise ValueError(f"img_channels <= 16, latent_dim and every stage a multiple of 32 are required, got {img_channels}, {latent_dim}, {up}")
        self.img_channels, self.latent_dim, self.up_channels, self.expand = img_channels, latent_dim, up, expand_factor
        self.latent_shape = (1, 1)
        ld = latent_dim
        table = ([("layers.0.weight", (ld, ld, 2, 2), W_INIT), ("layers.0.bias", (ld,), ONES)] + act_entries("layers.1", activation) +
                 [("layers.2.weight", (ld, ld, 1, 1), W_INIT)] + bn_entries("layers.3", ld) + act_entries("layers.4", activation))
        convs, dws = {"layers.0": hip.CONVT_2X2_S2, "layers.2": hip.CONV_1X1}, []
        for i, co in enumerate(up):
            pre = f"layers.{5 + i}."
            table += up_entries(pre, ld if i == 0 else up[i - 1], co, expand_factor, activation)
            convs.update(up_convs(pre))
            dws.append(pre + "layers.3.depthwise.1")
        last = f"layers.{5 + len(up)}."
        table += (conv_entries(last + "0", up[-1], up[-1], 3) + act_entries(last + "1", activation) +
                  [(last + "2.weight", (img_channels, up[-1], 3, 3), W_INIT), (last + "2.bias", (img_channels,), ZEROS)])
        convs.update({last + "0": hip.CONV_3X3_S1, last + "2": hip.CONV_3X3_S1})
        self._last = last
        self._build(table, BlockSpec(convs, dws, expand_factor, drop_p, activation, 4), initializer, device, dtype)

    def _program(self, u, inputs):
        act = self._spec.act
        z = u.conv_unit("layers.0", inputs[0], act=act)
        z = u.conv_bn_unit("layers.2", z, "layers.3", act)
        for i in range(len(self.up_channels)):
            z = up_program(u, f"layers.{5 + i}.", z, act)
        z = u.conv_unit(self._last + "0", z, act=act)
        return u.head_unit(self._last + "2", z)

    def _site_sizes(self, shapes):
        n, h, w, _ = shapes[0]
        h, w = 2 * h, 2 * w
        out, cin = [], self.latent_dim
        for c in self.up_channels:
            h, w = 2 * h, 2 * w
            out += sir_sites(n, h, w, cin, self.expand)
            cin = c
        return out

    # the head writes NCHW fp32 itself and takes its gradient in that layout
    _from_kernel = staticmethod(lambda o: o)
    _grad_to_kernel = staticmethod(lambda g: g.float().contiguous())

    def forward(self, input: Tensor) -> Tensor:
        return self._run(input)

    @torch.no_grad()
    def sample(self):
        raise NotImplementedError          # as in the reference (base_ae.py:251-259)

    def generate(self):
        raise NotImplementedError
