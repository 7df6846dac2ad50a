"""Thin tensor-level wrappers over the C ABI: shape bookkeeping, output allocation, pointer passing."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch import Tensor

from . import hip

BF16 = torch.bfloat16


def conv_out_hw(kind: int, ih: int, iw: int):
    if kind in (hip.CONV_3X3_S1, hip.CONV_1X1):
        return ih, iw
    if kind == hip.CONV_3X3_S2:
        return (ih - 1) // 2 + 1, (iw - 1) // 2 + 1
    if kind in (hip.CONVT_2X2_S2, hip.DGRAD_3X3_S2):
        return 2 * ih, 2 * iw
    if kind == hip.DGRAD_CONVT_2X2:
        return ih // 2, iw // 2
    raise ValueError(kind)


def conv_forward(kind: int, x0: Tensor, w: Tensor, bias: Optional[Tensor], cout: int, *, x1: Optional[Tensor] = None,
                 in_add: Optional[Tensor] = None, in_mask: Optional[Tensor] = None, in_scale: float = 1.0, act: int = 0,
                 res: Optional[Tensor] = None, out_mask: Optional[Tensor] = None, out_scale: float = 1.0,
                 out: Optional[Tensor] = None, f32_nchw: bool = False, out_cs: Optional[int] = None) -> Tensor:
    """x0 (and x1): NHWC bf16.  w: packed bf16 [slices][CoutW][C0+C1].  Returns NHWC bf16 [N,OH,OW,out_cs]
    (or NCHW fp32 [N,cout,OH,OW] when f32_nchw)."""
    n, ih, iw, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[3]
    assert x0.dtype == BF16 and x0.is_contiguous() and w.dtype == BF16 and w.is_contiguous()
    assert w.shape[2] == c0 + c1, (tuple(w.shape), c0, c1)
    oh, ow = conv_out_hw(kind, ih, iw)
    if out is None:
        if f32_nchw:
            out = torch.empty(n, cout, oh, ow, device=x0.device, dtype=torch.float32)
        else:
            out = torch.empty(n, oh, ow, out_cs or cout, device=x0.device, dtype=BF16)
    d = hip.ConvDesc(kind=kind, N=n, IH=ih, IW=iw, in0=hip.ptr(x0), in1=hip.ptr(x1), C0=c0, C1=c1, in_add=hip.ptr(in_add),
                     in_mask=hip.ptr(in_mask), in_mask_scale=in_scale, w=hip.ptr(w), bias=hip.ptr(bias), out=hip.ptr(out),
                     res=hip.ptr(res), out_mask=hip.ptr(out_mask), out_mask_scale=out_scale, Cout=cout, CoutW=w.shape[1],
                     out_cs=(0 if f32_nchw else out.shape[3]), act=act, out_mode=(hip.OUT_F32_NCHW if f32_nchw else hip.OUT_BF16_NHWC))
    hip.call("ddpm_conv_forward", C.byref(d), hip.stream())
    return out
