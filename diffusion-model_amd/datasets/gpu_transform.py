"""The reference's per-image dataset transforms, run on whole uint8 batches on the GPU.

The reference builds `transforms.Compose([Resize(img_size), ToTensor(), Lambda(x*2-1), Lambda(expand to 3 channels)])`
per dataset (datasets/mnist.py:26-39, datasets/fashion_mnist.py, datasets/flowers102.py:28-38 there) and applies it image
by image in DataLoader workers; `augmentation/augmentation.py:13` adds a random horizontal flip.  At > 10^4 images/s per
GPU those workers are the bottleneck, so the same arithmetic runs as one kernel on a uint8 batch that is already on
the device: PIL's bilinear resize (horizontal then vertical pass, 22-bit fixed-point coefficients, each pass rounded to 8
bits), /255, *2-1, channel replication, optional flip.  The coefficient tables are computed here exactly as PIL's
precompute_coeffs does (double precision), so the output is bit-identical to the CPU pipeline."""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from ..runtime import hip

_PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_tables(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """(coefficients int32 [out][ksize], bounds int32 [out][2] = (first source index, count), ksize) of PIL's
    ImagingResample for the BILINEAR filter (support 1), 8 bits per channel."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w = 1.0 - a if a < 1.0 else 0.0
            k[x] = w
            ww += w
        if ww != 0.0:
            k[:xmax] /= ww
        for x in range(ksize):
            v = k[x] * (1 << _PRECISION_BITS)
            coef[xx, x] = int(-0.5 + v) if k[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return coef, bounds, ksize


class GpuImageTransform:
    """uint8 batch [N, H0, W0] or [N, H0, W0, C0] on the GPU -> fp32 [N, channels, H, W] in [-1, 1]."""

    def __init__(self, img_size, channels: int = 3, flip_p: float = 0.0):
        self.size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        self.channels = int(channels)
        self.flip_p = float(flip_p)
        self._tables = {}

    def _axis(self, n_in: int, n_out: int, device):
        key = (n_in, n_out, str(device))
        if key not in self._tables:
            if n_in == n_out:
                self._tables[key] = (None, None, 0)       # PIL returns a copy when the size already matches
            else:
                c, b, k = pil_bilinear_tables(n_in, n_out)
                self._tables[key] = (torch.from_numpy(c).to(device), torch.from_numpy(b).to(device), k)
        return self._tables[key]

    def __call__(self, images: Tensor, flip: Optional[Tensor] = None) -> Tensor:
        if not images.is_cuda or images.dtype != torch.uint8:
            raise RuntimeError("GpuImageTransform takes uint8 image batches that are already on an MI355X (cuda) device")
        if images.dim() == 3:
            images = images.unsqueeze(-1)
        images = images.contiguous()
        n, h0, w0, c0 = images.shape
        h, w = self.size
        if h0 != h or w0 != w:
            # PIL resamples horizontally first, and only when that axis changes
            pass
        hc, hb, hk = self._axis(w0, w, images.device)
        vc, vb, vk = self._axis(h0, h, images.device)
        if flip is None and self.flip_p > 0.0:
            flip = torch.rand(n, device=images.device) < self.flip_p
        if flip is not None:
            flip = flip.to(torch.uint8).contiguous()
        out = torch.empty(n, self.channels, h, w, device=images.device, dtype=torch.float32)
        hip.call("ddpm_prepare_images", hip.ptr(images), n, h0, w0, c0, hip.ptr(out), self.channels, h, w, hip.ptr(hc), hip.ptr(hb), hk,
                 hip.ptr(vc), hip.ptr(vb), vk, hip.ptr(flip), hip.stream())
        return out
