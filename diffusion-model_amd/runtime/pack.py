"""Weight repacking between the torch parameter layouts the reference stores (fp32 OIHW / IOHW / out x in)
and the bf16 [slice][rows][K] layout the MFMA convolution kernels read.

The packed tensors are derived caches: the nn.Parameters stay fp32 in torch layout, so state_dicts
interchange with the reference (base_trainer.py:118-130).
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import hip


def _pad(w: Tensor, rows_to: int, k_to: int) -> Tensor:
    s, r, k = w.shape
    if r == rows_to and k == k_to:
        return w.contiguous()
    out = torch.zeros(s, rows_to, k_to, dtype=w.dtype, device=w.device)
    out[:, :r, :k] = w
    return out


def _ru(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def pack_fwd(kind: int, w: Tensor) -> Tensor:
    """bf16 [slices][CoutW][CinP] for the forward kinds."""
    if kind in (hip.CONV_3X3_S1, hip.CONV_3X3_S2):
        co, ci = w.shape[:2]
        p = w.permute(2, 3, 0, 1).reshape(9, co, ci)
    elif kind == hip.CONV_1X1:
        co, ci = w.shape[:2]
        p = w.reshape(1, co, ci)
    elif kind == hip.CONVT_2X2_S2:               # (Cin, Cout, 2, 2) -> slice a*2+b : [Cout][Cin]
        ci, co = w.shape[:2]
        p = w.permute(2, 3, 1, 0).reshape(4, co, ci)
    else:
        raise ValueError(kind)
    return _pad(p, _ru(co, 32), _ru(ci, 16)).to(torch.bfloat16)


def pack_dgrad(kind: int, w: Tensor) -> Tensor:
    """bf16 [slices][CinW][CoutP] for the data-gradient of a forward kind (rows = input channels)."""
    if kind == hip.CONV_3X3_S1:                  # run as CONV_3X3_S1 on dY with flipped taps
        co, ci = w.shape[:2]
        p = w.flip(2, 3).permute(2, 3, 1, 0).reshape(9, ci, co)
    elif kind == hip.CONV_3X3_S2:                # run as DGRAD_3X3_S2, slices indexed kh*3+kw
        co, ci = w.shape[:2]
        p = w.permute(2, 3, 1, 0).reshape(9, ci, co)
    elif kind == hip.CONV_1X1:
        co, ci = w.shape[:2]
        p = w.reshape(co, ci).t().reshape(1, ci, co)
    elif kind == hip.CONVT_2X2_S2:               # run as DGRAD_CONVT_2X2, slices a*2+b : [Cin][Cout]
        ci, co = w.shape[:2]
        p = w.permute(2, 3, 0, 1).reshape(4, ci, co)
    else:
        raise ValueError(kind)
    return _pad(p, _ru(ci, 32), _ru(co, 16)).to(torch.bfloat16)


DGRAD_KIND = {hip.CONV_3X3_S1: hip.CONV_3X3_S1, hip.CONV_3X3_S2: hip.DGRAD_3X3_S2, hip.CONV_1X1: hip.CONV_1X1,
              hip.CONVT_2X2_S2: hip.DGRAD_CONVT_2X2}
