"""CPU oracle of the reference's per-image dataset transforms (TEST INFRASTRUCTURE, not product code).

Restates datasets/mnist.py:9-14,26-39 and datasets/flowers102.py:28-38 of the reference with the libraries those files
call into: torchvision's transforms.Resize on a PIL image is PIL's Image.resize(BILINEAR); ToTensor is uint8 -> float32 / 255
in CHW order; then x * 2 - 1 and expand(3, -1, -1).  augmentation/augmentation.py:13's RandomHorizontalFlip is a flip of
the last axis.  torchvision itself is not installed in this image; PIL is, and it is the arithmetic that matters here.

Parity status: pinned to PIL (the reference's own resampling code path), not to stored fixtures: the reference holds no
fixture for its input pipeline."""
from __future__ import annotations

import numpy as np
import torch
from PIL import Image


def reference_transform(img: np.ndarray, size, channels: int = 3, flip: bool = False) -> torch.Tensor:
    """img: uint8 [H0, W0] or [H0, W0, 3] -> float32 [channels, H, W] in [-1, 1]"""
    h, w = (size, size) if isinstance(size, int) else size
    pil = Image.fromarray(img)
    if pil.size != (w, h):
        pil = pil.resize((w, h), Image.BILINEAR)            # transforms.Resize(img_size)
    arr = np.asarray(pil)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(arr.copy()).permute(2, 0, 1).to(torch.float32).div(255)   # ToTensor
    t = t * 2 - 1                                           # rescaling
    if t.shape[0] == 1 and channels > 1:
        t = t.expand(channels, -1, -1)                      # duplicate_channels
    if flip:
        t = t.flip(-1)                                      # RandomHorizontalFlip, when it fires
    return t.contiguous()
