"""Platform-independent pseudo-random tensors for fixtures (TEST INFRASTRUCTURE).

Everything is integer hashing of the element index followed by exact integer->double
arithmetic, so the same (shape, seed) gives bit-identical fp32 tensors on any machine
and any torch version; fixtures therefore never need to store weights or noise.
"""
from __future__ import annotations

from typing import Sequence

import torch
from torch import Tensor

_M = 0xFFFFFFFF


def _hash_u32(n: int, seed: int) -> Tensor:
    idx = torch.arange(n, dtype=torch.int64)
    h = (idx * 2654435761 + seed * 40503 + 12345) & _M
    h = ((h ^ (h >> 15)) * 2246822519) & _M
    h = ((h ^ (h >> 13)) * 3266489917) & _M
    return h ^ (h >> 16)


def _numel(shape: Sequence[int]) -> int:
    n = 1
    for d in shape:
        n *= int(d)
    return n


def hash_uniform(shape: Sequence[int], seed: int) -> Tensor:
    """fp32 tensor, uniform on [0, 1)."""
    n = _numel(shape)
    return (_hash_u32(n, seed).double() / 4294967296.0).float().view(*shape)


def hash_normal(shape: Sequence[int], seed: int) -> Tensor:
    """fp32 tensor, approximately N(0,1): centred Irwin-Hall sum of four hashed uniforms
    (variance 4/12, rescaled), support +-3.46 sigma."""
    n = _numel(shape)
    acc = torch.zeros(n, dtype=torch.float64)
    for k in range(4):
        acc += _hash_u32(n, seed * 7919 + 104729 * (k + 1)).double() / 4294967296.0
    return ((acc - 2.0) * (3.0 ** 0.5)).float().view(*shape)
