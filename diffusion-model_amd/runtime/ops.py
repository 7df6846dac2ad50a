"""Thin tensor-level wrappers over the C ABI: shape bookkeeping, output allocation, pointer passing."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch
from torch import Tensor

from . import hip

BF16 = torch.bfloat16

# Launch probe (runtime.roofline installs one): an object whose .launch(kernel_name, cost_fn, fn) runs fn() between
# two timing events on the launch stream; cost_fn() -> (algorithmic bytes, flops, shape tag) of that launch.  None in normal use.
PROBE = None


def _launch(kernel: str, cost, fn):
    if PROBE is None:
        fn()
    else:
        PROBE.launch(kernel, cost, fn)


def _nb(*tensors) -> int:
    return sum(t.numel() * t.element_size() for t in tensors if t is not None)


# MACs per output pixel and (input channel x output channel) pair
_TAPS = {hip.CONV_3X3_S1: 9.0, hip.CONV_3X3_S2: 9.0, hip.CONV_1X1: 1.0, hip.CONVT_2X2_S2: 1.0, hip.DGRAD_3X3_S2: 2.25,
         hip.DGRAD_CONVT_2X2: 4.0}
_SLICES = {hip.CONV_3X3_S1: 9, hip.CONV_3X3_S2: 9, hip.CONV_1X1: 1, hip.CONVT_2X2_S2: 4, hip.DGRAD_3X3_S2: 9, hip.DGRAD_CONVT_2X2: 4}


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
                 out: Optional[Tensor] = None, f32_nchw: bool = False, out_cs: Optional[int] = None,
                 row0: int = 0, rows: Optional[int] = None, sum_out: Optional[Tensor] = None, gn=None, c0_used: int = 0, act2: Optional[int] = None, gnb=None,
                 gn_part_groups: int = 0, in_affine=None, keep_raw: bool = True):
    """x0 (and x1): NHWC bf16.  w: packed bf16 [slices][CoutW][C0+C1].  Returns NHWC bf16 [N,OH,OW,out_cs]
    (or NCHW fp32 [N,cout,OH,OW] when f32_nchw).  row0/rows select a sub-range of weight rows (= output channels).
    sum_out (fp32 [N][cout], zeroed by the caller) receives per-image spatial sums of the result before mask/res.
    gnb = (raw, gamma, beta, groups, act, stats, part, draw): x0 is the gradient of a conv -> GroupNorm -> act unit's output;
    the GroupNorm/activation backward runs in the loader (where conv_gnb_fusable says so), filling part [N][3][C] and draw.
    act2: also return act2(result) as a second tensor (the result itself without activation): returns (raw, y).
    c0_used: promise that only the first c0_used (1..4) channels of x0 are non-zero (zero-padded image tensors).
    gn = (gamma, beta, groups, act, res or None[, the scalar of swish / hardswish]): GroupNorm + activation (+ residual) finished in the same launch (only
    where conv_gn_fusable says so); returns (raw, y, stats) instead of the raw result alone; keep_raw=False (inference): the raw
    result is not stored and comes back as None.
    gn_part_groups > 0: returns (raw, part) with part = fp32 [N][groups][tiles][2] per-tile sums (x, x^2) of the result for
    gn_act_forward(stat_part=part), or None where the launch cannot deliver them (small images, odd shapes).
    in_affine = (affine, act): x0 is the raw input of a GroupNorm + activation that has not been applied; affine = fp32
    [N][C0][2] from gn_fold_affine, act = none / gelu / silu: the loader normalises and activates (inference on large images)."""
    n, ih, iw, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[3]
    assert x0.dtype == BF16 and x0.is_contiguous() and w.dtype == BF16 and w.is_contiguous()
    assert w.shape[2] == c0 + c1, (tuple(w.shape), c0, c1)
    oh, ow = conv_out_hw(kind, ih, iw)
    no_raw = gn is not None and not keep_raw and out is None and not f32_nchw
    if no_raw:
        pass
    elif out is None:
        if f32_nchw:
            out = torch.empty(n, cout, oh, ow, device=x0.device, dtype=torch.float32)
        else:
            out = torch.empty(n, oh, ow, out_cs or cout, device=x0.device, dtype=BF16)
    d = hip.ConvDesc(kind=kind, N=n, IH=ih, IW=iw, in0=hip.ptr(x0), in1=hip.ptr(x1), C0=c0, C1=c1, in_add=hip.ptr(in_add),
                     in_mask=hip.ptr(in_mask), in_mask_scale=in_scale, w=w.data_ptr() + row0 * w.shape[2] * 2, bias=hip.ptr(bias), out=hip.ptr(out),
                     res=hip.ptr(res), out_mask=hip.ptr(out_mask), out_mask_scale=out_scale, Cout=cout,
                     CoutW=(w.shape[1] if rows is None else (rows + 31) // 32 * 32), out_cs=(0 if f32_nchw else (out_cs or cout) if no_raw else out.shape[3]),
                     w_rows=w.shape[1], act=act, out_mode=(hip.OUT_F32_NCHW if f32_nchw else hip.OUT_BF16_NHWC),
                     sum_out=hip.ptr(sum_out), C0_used=c0_used)
    y = stats = part = None
    if in_affine is not None:
        aff, aact = in_affine
        assert aff.dtype == torch.float32 and aff.is_contiguous() and aff.shape == (n, c0, 2) and x1 is None and in_add is None
        d.in_affine, d.in_affine_act = hip.ptr(aff), aact
    if gn_part_groups > 0:
        d.gn_part_groups = gn_part_groups
        tiles = hip.lib().ddpm_conv_gn_part_tiles(C.byref(d))
        if tiles > 0:
            part = torch.empty(n, gn_part_groups, tiles, 2, device=x0.device, dtype=torch.float32)
            d.gn_part = hip.ptr(part)
        else:
            d.gn_part_groups = 0
    if gnb is not None:
        g_raw, g_gamma, g_beta, g_groups, g_act, g_stats, g_part, g_draw = gnb
        d.gnb_raw, d.gnb_gamma, d.gnb_beta, d.gnb_stats = hip.ptr(g_raw), hip.ptr(g_gamma), hip.ptr(g_beta), hip.ptr(g_stats)
        d.gnb_part, d.gnb_draw, d.gnb_groups, d.gnb_act = hip.ptr(g_part), hip.ptr(g_draw), g_groups, g_act
    if act2 is not None:
        y = torch.empty_like(out)
        d.out2, d.act2 = hip.ptr(y), act2
    if gn is not None:
        gamma, beta, groups, gact, gres = gn[:5]
        d.gn_act_beta = _scalar_ptr(gact, gn[5] if len(gn) > 5 else None)
        y = torch.empty(n, oh, ow, cout, device=x0.device, dtype=BF16)
        stats = torch.empty(n, groups, 2, device=x0.device, dtype=torch.float32)
        d.gn_gamma, d.gn_beta, d.gn_out, d.gn_stats, d.gn_res = hip.ptr(gamma), hip.ptr(beta), hip.ptr(y), hip.ptr(stats), hip.ptr(gres)
        d.gn_groups, d.gn_act, d.gn_eps = groups, gact, 1e-5

    def cost():
        # every operand once: activations in, weight rows used, bias, epilogue operands, result
        wbytes = _SLICES[kind] * d.CoutW * (c0 + c1) * 2
        obytes = 0 if no_raw else n * oh * ow * cout * (4 if f32_nchw else 2)
        extra = 0 if gnb is None else _nb(gnb[0], gnb[5], gnb[6], gnb[7])
        return (_nb(x0, x1, in_add, in_mask, res, out_mask, sum_out, y, stats, None if gn is None else gn[4]) + wbytes + cout * 4 + obytes + extra,
                2.0 * n * oh * ow * cout * (c0 + c1) * _TAPS[kind], f"k{kind} {ih}x{iw} {c0}+{c1}->{cout}")

    # which of the convolution kernels takes the launch is the library's choice: the probe asks after the call
    _launch(lambda: hip.CONV_KERNEL_NAMES[hip.lib().ddpm_conv_last_kernel()], cost, lambda: hip.call("ddpm_conv_forward", C.byref(d), hip.stream()))
    if act2 is not None:
        return out, y
    if gn_part_groups > 0:
        return out, part
    return out if gn is None else (out, y, stats)


def conv_pipe_wins(kind: int, x0: Tensor, x1: Optional[Tensor], cout: int) -> bool:
    """Whether a plain conv_forward of this problem runs on the pipelined 256 x 128 tile kernel (csrc/conv_pipe.hip) by the
    library's own size rule: 3x3 stride 1 on 4x4 / 8x8 / 16x16 images, channel counts in its granules, and at least one
    tile per CU."""
    n, ih, iw, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[3]
    if kind != hip.CONV_3X3_S1 or ih != iw or iw not in (4, 8, 16) or c0 % 64 or c1 % 64 or cout % 128 or os.environ.get("DDPM_CONV_PIPE", "1") == "0":
        return False
    return n * ih * iw * (cout // 128) >= 256 * 256 and (c0 + c1) * 9 >= 512


def conv_gnb_fusable(kind: int, dy: Tensor, cout: int, groups: int, masked: bool = False) -> bool:
    """Whether conv_forward(kind, dy, ..., gnb=...) is available: kind = the data-gradient kind, dy = gradient of the unit's
    output ([N,H,W,C]), cout = channels of the data-gradient, groups = the unit's GroupNorm groups."""
    n, ih, iw, c0 = dy.shape
    d = hip.ConvDesc(kind=kind, N=n, IH=ih, IW=iw, C0=c0, C1=0, Cout=cout, out_cs=cout, CoutW=(cout + 31) // 32 * 32,
                     w_rows=(cout + 31) // 32 * 32, out_mode=hip.OUT_BF16_NHWC, gnb_groups=groups,
                     out_mask=(1 if masked else None))
    return bool(hip.lib().ddpm_conv_gnb_fusable(C.byref(d)))


def conv_gn_fusable(kind: int, x0: Tensor, x1: Optional[Tensor], cout: int, groups: int) -> bool:
    """Whether conv_forward(..., gn=...) is available for this convolution (no masks, no conv-side activation / residual)."""
    n, ih, iw, c0 = x0.shape
    d = hip.ConvDesc(kind=kind, N=n, IH=ih, IW=iw, C0=c0, C1=0 if x1 is None else x1.shape[3], Cout=cout, out_cs=cout,
                     CoutW=(cout + 31) // 32 * 32, w_rows=(cout + 31) // 32 * 32, out_mode=hip.OUT_BF16_NHWC, gn_groups=groups)
    return bool(hip.lib().ddpm_conv_gn_fusable(C.byref(d)))


def wgrad_desc(kind, x0, dy, cout, dw, part, x1, in_add, in_mask, in_scale, tiles_per_split=0):
    """ddpm_wgrad_desc of one weight-gradient problem (the caller keeps the tensors alive)"""
    n, ih, iw, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[3]
    return hip.WgradDesc(kind=kind, N=n, IH=ih, IW=iw, in0=hip.ptr(x0), in1=hip.ptr(x1), C0=c0, C1=c1, in_add=hip.ptr(in_add),
                         in_mask=hip.ptr(in_mask), in_mask_scale=in_scale, dy=hip.ptr(dy), Cout=cout, dy_cs=dy.shape[3], dw=hip.ptr(dw),
                         part=hip.ptr(part), tiles_per_split=tiles_per_split)


def conv_wgrad_splits(kind: int, x0: Tensor, dy: Tensor, cout: int, *, x1: Optional[Tensor] = None, tiles_per_split: int = 0) -> int:
    """Number of partial-sum rows conv_wgrad(part=...) writes for this problem."""
    d = wgrad_desc(kind, x0, dy, cout, None, None, x1, None, None, 1.0, tiles_per_split)
    r = hip.lib().ddpm_conv_wgrad_splits(C.byref(d))
    if r < 0:
        raise RuntimeError(hip.lib().ddpm_last_error().decode())
    return r


def conv_wgrad_pipe_splits(kind: int, x0: Tensor, dy: Tensor, cout: int, *, x1: Optional[Tensor] = None, in_add=None, in_mask=None) -> int:
    """Rows of `part` when a standalone conv_wgrad (tiles_per_split = 0) would run on the pipelined kernel for deep wide 3x3
    convolutions at large batch; 0 when it would not (shape, or too small to pay)."""
    if kind != hip.CONV_3X3_S1:
        return 0
    d = wgrad_desc(kind, x0, dy, cout, None, None, x1, in_add, in_mask, 1.0, 0)
    return int(hip.lib().ddpm_conv_wgrad_pipe_splits(C.byref(d)))


def wgrad_cost(kind, x0, dy, cout, x1=None):
    """() -> (algorithmic bytes, flops, shape tag) of one weight-gradient problem"""
    n, ih, iw, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[3]
    oh, ow = dy.shape[1], dy.shape[2]
    return lambda: (_nb(x0, x1, dy) + 4 * _SLICES[kind] * cout * (c0 + c1), 2.0 * n * oh * ow * cout * (c0 + c1) * _TAPS[kind],
                    f"k{kind} {ih}x{iw} {c0}+{c1}->{cout}")


def conv_wgrad(kind: int, x0: Tensor, dy: Tensor, cout: int, dw: Optional[Tensor], *, x1: Optional[Tensor] = None,
               in_add: Optional[Tensor] = None, in_mask: Optional[Tensor] = None, in_scale: float = 1.0,
               part: Optional[Tensor] = None, tiles_per_split: int = 0) -> Optional[Tensor]:
    """Weight gradient.  x0/x1: the forward inputs; dy: NHWC bf16.  Either dw (fp32 [slices][cout][C0+C1], kernel layout)
    += gradient with atomics, or - when `part` (fp32 [splits][slices][cout][C0+C1]) is given - every position split stores
    its partial sums there for a later reduce_rows pass."""
    n, ih, iw, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[3]
    assert dy.dtype == BF16 and dy.is_contiguous()
    if part is None:
        assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.shape[1] == cout and dw.shape[2] == c0 + c1, (tuple(dw.shape), cout, c0, c1)
    d = wgrad_desc(kind, x0, dy, cout, dw, part, x1, in_add, in_mask, in_scale, tiles_per_split)
    oh, ow = dy.shape[1], dy.shape[2]
    _launch("conv_wgrad_kernel", lambda: (_nb(x0, x1, in_add, in_mask, dy) + 4 * _SLICES[kind] * cout * (c0 + c1),
                                          2.0 * n * oh * ow * cout * (c0 + c1) * _TAPS[kind], f"k{kind} {ih}x{iw} {c0}+{c1}->{cout}"),
            lambda: hip.call("ddpm_conv_wgrad", C.byref(d), hip.stream()))
    return dw


def _gn_scratch(d, backward: int, device) -> Optional[Tensor]:
    """workspace for the many-workgroup path of large images (None when the shape runs on one workgroup per slab)"""
    need = hip.lib().ddpm_gn_scratch_floats(C.byref(d), backward)
    if need < 0:
        raise RuntimeError(hip.lib().ddpm_last_error().decode())
    if need == 0:
        return None
    t = torch.empty(need, device=device, dtype=torch.float32)
    d.scratch, d.scratch_floats = t.data_ptr(), need
    return t


def gn_act_forward(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], groups: int, act: int, *,
                   res: Optional[Tensor] = None, mask: Optional[Tensor] = None, mask_scale: float = 1.0,
                   stats: Optional[Tensor] = None, out: Optional[Tensor] = None, eps: float = 1e-5, act_beta: Optional[Tensor] = None,
                   fixed_stats: bool = False, stat_part: Optional[Tensor] = None):
    """x: NHWC bf16.  Returns (y, stats) with stats fp32 [N][G][2] (None when groups == 0).  act_beta: the activation's
    learnable scalar (a device fp32 tensor with one element), required for swish / hardswish.  fixed_stats: normalise with
    the (mean, rstd) pairs given in `stats` instead of computing them.  stat_part: fp32 [N][G][parts][2] partial sums of x
    from the convolution that produced it (conv_forward(gn_part_groups=...)); large images then skip their statistics pass."""
    n, h, w, c = x.shape
    assert x.dtype == BF16 and x.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    if groups > 0 and stats is None:
        stats = torch.empty(n, groups, 2, device=x.device, dtype=torch.float32)
    d = hip.GnDesc(x=hip.ptr(x), gamma=hip.ptr(gamma), beta=hip.ptr(beta), res=hip.ptr(res), mask=hip.ptr(mask), mask_scale=mask_scale,
                   y=hip.ptr(out), stats=hip.ptr(stats), dy=None, part=None, N=n, HW=h * w, C=c, G=groups,
                   act=act, eps=eps, act_beta=_scalar_ptr(act, act_beta), fixed_stats=int(fixed_stats))
    if stat_part is not None:
        assert stat_part.dtype == torch.float32 and stat_part.is_contiguous() and stat_part.shape[:2] == (n, groups) and stat_part.shape[3] == 2
        d.stat_part, d.stat_parts = hip.ptr(stat_part), stat_part.shape[2]
    scratch = _gn_scratch(d, 0, x.device)
    _launch("gn_fwd_split_kernel" if scratch is not None else "gn_fwd_kernel", lambda: (_nb(x, res, mask, out, gamma, beta, stats), 0.0, f"{h}x{w} C{c} G{groups} a{act}"),
            lambda: hip.call("ddpm_gn_act_forward", C.byref(d), hip.stream()))
    return out, stats


def gn_fold_affine(stat_part: Tensor, gamma: Tensor, beta: Tensor, hw: int, eps: float = 1e-5):
    """stat_part: fp32 [N][G][parts][2] partial sums from the producing convolution.  Returns (stats [N][G][2], affine [N][C][2]):
    the GroupNorm's statistics and the per-image (scale, shift) of every channel, for conv_forward(in_affine=...)."""
    n, g, parts, _ = stat_part.shape
    c = gamma.numel()
    stats = torch.empty(n, g, 2, device=stat_part.device, dtype=torch.float32)
    affine = torch.empty(n, c, 2, device=stat_part.device, dtype=torch.float32)
    _launch("gn_fold_affine_kernel", lambda: (_nb(stat_part, stats, affine, gamma, beta), 0.0, f"N{n} C{c} G{g} parts{parts}"),
            lambda: hip.call("ddpm_gn_fold_affine", hip.ptr(stat_part), parts, hip.ptr(gamma), hip.ptr(beta), n, hw, c, g, float(eps),
                             hip.ptr(stats), hip.ptr(affine), hip.stream()))
    return stats, affine


def gn_act_backward(x: Tensor, dy: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], groups: int, act: int,
                    stats: Optional[Tensor], part: Tensor, *, mask: Optional[Tensor] = None, mask_scale: float = 1.0,
                    out: Optional[Tensor] = None, eps: float = 1e-5, act_beta: Optional[Tensor] = None,
                    dbeta_part: Optional[Tensor] = None, fixed_stats: bool = False) -> Tensor:
    """Returns dx.  part: fp32 [N][3][C] receives per-image partial sums (dgamma, dbeta, bias gradient).  swish / hardswish:
    act_beta = the activation's scalar, dbeta_part = fp32 [N][C] that receives the partial sums of the scalar's gradient."""
    n, h, w, c = x.shape
    assert dy.shape == x.shape and dy.dtype == BF16 and dy.is_contiguous()
    assert part.dtype == torch.float32 and part.numel() == n * 3 * c
    if out is None:
        out = torch.empty_like(x)
    d = hip.GnDesc(x=hip.ptr(x), gamma=hip.ptr(gamma), beta=hip.ptr(beta), res=None, mask=hip.ptr(mask), mask_scale=mask_scale,
                   y=hip.ptr(out), stats=hip.ptr(stats), dy=hip.ptr(dy), part=hip.ptr(part), N=n, HW=h * w, C=c, G=groups, act=act,
                   eps=eps, act_beta=_scalar_ptr(act, act_beta), dbeta_part=hip.ptr(dbeta_part), fixed_stats=int(fixed_stats))
    assert act not in hip.ACT_WITH_SCALAR or (dbeta_part is not None and dbeta_part.numel() == n * c and dbeta_part.dtype == torch.float32)
    scratch = _gn_scratch(d, 1, x.device)
    _launch("gn_bwd_split_kernel" if scratch is not None else "gn_bwd_kernel", lambda: (_nb(x, dy, mask, out, gamma, beta, stats, part), 0.0, f"{h}x{w} C{c} G{groups} a{act}"),
            lambda: hip.call("ddpm_gn_act_backward", C.byref(d), hip.stream()))
    return out


def batchnorm_act_forward(x: Tensor, gamma: Tensor, beta: Tensor, act: int, *, training: bool, running_mean: Tensor, running_var: Tensor,
                          num_batches: Optional[Tensor] = None, momentum: float = 0.1, eps: float = 1e-5, act_beta: Optional[Tensor] = None):
    """act(BatchNorm2d(x)) on NHWC bf16 (torch.nn.BatchNorm2d semantics).  Training: batch statistics per channel over
    N*H*W - the GroupNorm kernels on the tensor described as one image of N*H*W pixels with one group per channel - and
    the running statistics advance; eval: the running statistics are the (fixed) statistics.  Returns (y, stats [1][C][2])."""
    n, h, w, c = x.shape
    xv = x.view(1, n * h * w, 1, c)
    if training:
        y, stats = gn_act_forward(xv, gamma, beta, c, act, eps=eps, act_beta=act_beta)
        hip.call("ddpm_bn_update_running", hip.ptr(stats), hip.ptr(running_mean), hip.ptr(running_var), hip.ptr(num_batches), c, n * h * w,
                 float(momentum), float(eps), hip.stream())
    else:
        stats = torch.empty(1, c, 2, device=x.device, dtype=torch.float32)
        hip.call("ddpm_bn_eval_stats", hip.ptr(running_mean), hip.ptr(running_var), hip.ptr(stats), c, float(eps), hip.stream())
        y, _ = gn_act_forward(xv, gamma, beta, c, act, eps=eps, act_beta=act_beta, stats=stats, fixed_stats=True)
    return y.view(n, h, w, c), stats


def batchnorm_act_backward(x: Tensor, dy: Tensor, gamma: Tensor, beta: Tensor, act: int, stats: Tensor, part: Tensor, *, training: bool,
                           eps: float = 1e-5, act_beta: Optional[Tensor] = None, dbeta_part: Optional[Tensor] = None) -> Tensor:
    """Gradient w.r.t. x of the above.  part: fp32 [3][C] receives d gamma, d beta and the column sums of dx (complete sums: the
    whole batch is one 'image' here); dbeta_part: fp32 [C] for the swish / hardswish scalar."""
    n, h, w, c = x.shape
    dx = gn_act_backward(x.view(1, n * h * w, 1, c), dy.view(1, n * h * w, 1, c), gamma, beta, c, act, stats, part, eps=eps, act_beta=act_beta,
                         dbeta_part=dbeta_part, fixed_stats=not training)
    return dx.view(n, h, w, c)


def _scalar_ptr(act: int, beta: Optional[Tensor]):
    """device pointer of an activation's learnable scalar; swish / hardswish must bring one"""
    if act not in hip.ACT_WITH_SCALAR:
        return None
    if beta is None or beta.numel() != 1 or beta.dtype != torch.float32 or not beta.is_cuda:
        raise ValueError("swish / hardswish need their learnable scalar as a one-element fp32 device tensor")
    return hip.ptr(beta)


def act_backward_splits(x: Tensor) -> int:
    """Pixel ranges per image for act_backward (0: shape not supported, use gn_act_backward with groups = 0)."""
    n, h, w, c = x.shape
    if c not in (8, 16, 32, 64):
        return 0
    p = 1
    while (h * w) % (2 * p) == 0 and h * w * c // (8 * p) > 1024 and p < 64:      # ~4 vectors per thread
        p *= 2
    return p


def act_backward(x: Tensor, dy: Tensor, act: int, part: Tensor, splits: int) -> Tensor:
    """dx = dy * act'(x); part: fp32 [N * splits][C] receives the column sums of dx per pixel range."""
    n, h, w, c = x.shape
    assert dy.shape == x.shape and dy.dtype == BF16 and dy.is_contiguous() and x.is_contiguous()
    assert part.dtype == torch.float32 and part.numel() == n * splits * c
    out = torch.empty_like(x)
    _launch("act_bwd_kernel", lambda: (_nb(x, dy, out, part), 0.0, f"{h}x{w} C{c} a{act}"),
            lambda: hip.call("ddpm_act_backward", hip.ptr(x), hip.ptr(dy), hip.ptr(out), hip.ptr(part), n, h * w, c, splits, act, hip.stream()))
    return out


def _ptr_array(tensors):
    arr = (C.c_void_p * hip.EMBED_MAX)()
    for i, t in enumerate(tensors):
        arr[i] = hip.ptr(t)
    return arr


def embed_forward(sin: Tensor, w0s, b0s, w1s, b1s, act: int, record: bool):
    """All time-embedding MLPs in one launch.  sin: bf16 [N,1,1,32]; w0s[i]: bf16 [1][2C][32], w1s[i]: bf16 [1][C][2C].
    Returns (outs: list of fp32 [N][C], saved: list of (raw0, h) bf16 [N,1,1,2C] or None)."""
    n, t_dim = sin.shape[0], sin.shape[-1]
    k = len(w0s)
    cs = [b.shape[0] for b in b1s]
    assert k <= hip.EMBED_MAX and all(w.shape[-2:] == (2 * c, t_dim) for w, c in zip(w0s, cs)) and all(w.shape[-2:] == (c, 2 * c) for w, c in zip(w1s, cs))
    outs = [torch.empty(n, c, device=sin.device, dtype=torch.float32) for c in cs]
    saved = [(torch.empty(n, 1, 1, 2 * c, device=sin.device, dtype=BF16), torch.empty(n, 1, 1, 2 * c, device=sin.device, dtype=BF16))
             for c in cs] if record else None
    d = hip.EmbedDesc(sin=hip.ptr(sin), N=n, t_dim=t_dim, act=act, nmlp=k)
    for i, c in enumerate(cs):
        d.C[i] = c
    d.w0, d.b0, d.w1, d.b1, d.out = _ptr_array(w0s), _ptr_array(b0s), _ptr_array(w1s), _ptr_array(b1s), _ptr_array(outs)
    if record:
        d.raw0, d.h = _ptr_array([s[0] for s in saved]), _ptr_array([s[1] for s in saved])
    _launch("embed_fwd_kernel", lambda: (_nb(sin, *w0s, *w1s, *outs) + (sum(_nb(*s) for s in saved) if saved else 0),
                                         sum(2.0 * n * (t_dim * 2 * c + 2 * c * c) for c in cs), f"{k} MLPs"),
            lambda: hip.call("ddpm_embed_forward", C.byref(d), hip.stream()))
    return outs, saved


def embed_backward(dtembs, raw0s, w1ds, act: int, db1s, db0s):
    """Backward data path of all MLPs in one launch.  Returns (dt16s: bf16 [N,1,1,C], dzs: bf16 [N,1,1,2C])."""
    n = dtembs[0].shape[0]
    cs = [g.shape[1] for g in dtembs]
    dev = dtembs[0].device
    dt16s = [torch.empty(n, 1, 1, c, device=dev, dtype=BF16) for c in cs]
    dzs = [torch.empty(n, 1, 1, 2 * c, device=dev, dtype=BF16) for c in cs]
    d = hip.EmbedBwdDesc(N=n, act=act, nmlp=len(cs), pad=0)
    for i, c in enumerate(cs):
        d.C[i] = c
        assert dtembs[i].dtype == torch.float32 and dtembs[i].is_contiguous() and w1ds[i].shape[-2:] == (2 * c, c)
        assert db1s[i].numel() == (n + 31) // 32 * c and db0s[i].numel() == (n + 31) // 32 * 2 * c
    d.dtemb, d.raw0, d.w1d, d.dt16, d.dz = _ptr_array(dtembs), _ptr_array(raw0s), _ptr_array(w1ds), _ptr_array(dt16s), _ptr_array(dzs)
    d.db1, d.db0 = _ptr_array(db1s), _ptr_array(db0s)
    _launch("embed_bwd_kernel", lambda: (_nb(*dtembs, *raw0s, *w1ds, *dt16s, *dzs), sum(4.0 * n * c * c for c in cs), f"{len(cs)} MLPs"),
            lambda: hip.call("ddpm_embed_backward", C.byref(d), hip.stream()))
    return dt16s, dzs


def sir_supported(c: int, h: int, w: int, e: int, groups: int, act: int) -> bool:
    return bool(hip.lib().ddpm_sir_supported(c, h, w, e, groups, act))


def sir_backward_supported(c: int, h: int, w: int, e: int, groups: int, act: int) -> bool:
    return bool(hip.lib().ddpm_sir_backward_supported(c, h, w, e, groups, act))


def sir_forward(x: Tensor, we: Tensor, be: Tensor, g1: Tensor, b1: Tensor, wd: Tensor, bd: Tensor, g2: Tensor, b2: Tensor,
                wp: Tensor, bp: Tensor, g3: Tensor, b3: Tensor, groups: int, act: int, *, masks=(None, None, None),
                mask_scale: float = 1.0, record: bool = False, eps: float = 1e-5, act_betas=(None, None)):
    """One fused SeparableInvertResidual block.  x: NHWC bf16 [N,H,W,C]; we: bf16 [1][E][C]; wd: fp32 [9][E]; wp: bf16 [1][C][E].
    act_betas: swish / hardswish, the learnable scalars of the block's two activation modules (one-element fp32 device tensors).
    Returns (y, saved) where saved is None or the dict of tensors the backward pass needs
    (raw_e, y_e, st1, raw_d, y_d, st2, raw_o, st3)."""
    n, h, w, c = x.shape
    e = wd.shape[1]
    assert x.dtype == BF16 and x.is_contiguous() and we.dtype == BF16 and wp.dtype == BF16 and wd.dtype == torch.float32
    assert we.shape[-2:] == (e, c) and wp.shape[-2:] == (c, e), (tuple(we.shape), tuple(wp.shape), e, c)
    y = torch.empty_like(x)
    saved = None
    if record:
        big = lambda: torch.empty(n, h, w, e, device=x.device, dtype=BF16)                        # noqa: E731
        st = lambda: torch.empty(n, groups, 2, device=x.device, dtype=torch.float32)            # noqa: E731
        saved = dict(raw_e=big(), y_e=big(), st1=st(), raw_d=big(), y_d=big(), st2=st(), raw_o=torch.empty_like(x), st3=st())
    sv = saved or {}
    d = hip.SirDesc(x=hip.ptr(x), y=hip.ptr(y), we=hip.ptr(we), be=hip.ptr(be), g1=hip.ptr(g1), b1=hip.ptr(b1), wd=hip.ptr(wd),
                    bd=hip.ptr(bd), g2=hip.ptr(g2), b2=hip.ptr(b2), wp=hip.ptr(wp), bp=hip.ptr(bp), g3=hip.ptr(g3), b3=hip.ptr(b3),
                    m_in=hip.ptr(masks[0]), m_e=hip.ptr(masks[1]), m_d=hip.ptr(masks[2]), mask_scale=mask_scale, eps=eps,
                    raw_e=hip.ptr(sv.get("raw_e")), y_e=hip.ptr(sv.get("y_e")), st1=hip.ptr(sv.get("st1")),
                    raw_d=hip.ptr(sv.get("raw_d")), y_d=hip.ptr(sv.get("y_d")), st2=hip.ptr(sv.get("st2")),
                    raw_o=hip.ptr(sv.get("raw_o")), st3=hip.ptr(sv.get("st3")), N=n, C=c, H=h, W=w, E=e, groups=groups, act=act, pad=0,
                    act_b1=_scalar_ptr(act, act_betas[0]), act_b2=_scalar_ptr(act, act_betas[1]))

    def cost():
        nb = _nb(x, y, we, wp, wd) + sum(_nb(t) for t in sv.values()) + sum(_nb(m) for m in masks)
        return nb, 2.0 * n * h * w * (2 * c * e + 9 * e), f"{h}x{w} C{c} E{e}" + (" rec" if record else "")

    _launch("sir_fwd_big_kernel" if h * w * e > 256 * 256 - 1 else "sir_fwd_kernel", cost, lambda: hip.call("ddpm_sir_forward", C.byref(d), hip.stream()))
    return y, saved


def sir_workgroups(hw: int, n: int) -> int:
    return hip.lib().ddpm_sir_workgroups(hw, n)


def sir_backward(dy: Tensor, saved: dict, g1: Tensor, b1: Tensor, g2: Tensor, b2: Tensor, g3: Tensor, wpd: Tensor, wed: Tensor,
                 wdf: Tensor, groups: int, act: int, part1: Tensor, part2: Tensor, part3: Tensor, dwpart: Tensor, *,
                 masks=(None, None, None), mask_scale: float = 1.0, act_betas=(None, None), dbeta_parts=(None, None)):
    """Data-gradient chain of one fused block.  saved: what sir_forward(record=True) returned.  wpd: bf16 [1][E][C],
    wed: bf16 [1][C][E] (data-gradient weight copies), wdf: fp32 [9][E] flipped taps.  swish / hardswish: act_betas = the two
    activation modules' scalars, dbeta_parts = two fp32 workspaces of sir_workgroups(h * w, n) * 8 partial sums of their
    gradients.  Returns (dx, d_o, d_e)."""
    n, h, w, c = dy.shape
    e = wdf.shape[1]
    assert dy.dtype == BF16 and dy.is_contiguous() and wpd.shape[-2:] == (e, c) and wed.shape[-2:] == (c, e)
    assert part1.numel() == n * 3 * e and part2.numel() == n * 3 * e and part3.numel() == n * 3 * c
    assert dwpart.numel() == sir_workgroups(h * w, n) * 10 * e
    dx = torch.empty_like(dy)
    d_o = torch.empty_like(dy)
    d_e = torch.empty(n, h, w, e, device=dy.device, dtype=BF16)
    d = hip.SirBwdDesc(dy=hip.ptr(dy), raw_e=hip.ptr(saved["raw_e"]), raw_d=hip.ptr(saved["raw_d"]), raw_o=hip.ptr(saved["raw_o"]),
                       y_e=hip.ptr(saved["y_e"]), st1=hip.ptr(saved["st1"]), st2=hip.ptr(saved["st2"]), st3=hip.ptr(saved["st3"]),
                       g1=hip.ptr(g1), b1=hip.ptr(b1), g2=hip.ptr(g2), b2=hip.ptr(b2), g3=hip.ptr(g3), wpd=hip.ptr(wpd), wed=hip.ptr(wed),
                       wdf=hip.ptr(wdf), m_in=hip.ptr(masks[0]), m_e=hip.ptr(masks[1]), m_d=hip.ptr(masks[2]), dx=hip.ptr(dx),
                       d_o=hip.ptr(d_o), d_e=hip.ptr(d_e), part1=hip.ptr(part1), part2=hip.ptr(part2), part3=hip.ptr(part3),
                       dwpart=hip.ptr(dwpart), mask_scale=mask_scale, N=n, C=c, H=h, W=w, E=e, groups=groups, act=act,
                       act_b1=_scalar_ptr(act, act_betas[0]), act_b2=_scalar_ptr(act, act_betas[1]))
    if act in hip.ACT_WITH_SCALAR:
        assert all(p is not None and p.dtype == torch.float32 and p.numel() == sir_workgroups(h * w, n) * 8 for p in dbeta_parts)
        d.dbeta1, d.dbeta2 = hip.ptr(dbeta_parts[0]), hip.ptr(dbeta_parts[1])

    def cost():
        nb = _nb(dy, dx, d_o, d_e, wpd, wed, saved["raw_e"], saved["raw_d"], saved["raw_o"], saved["y_e"]) + sum(_nb(m) for m in masks)
        return nb, 2.0 * n * h * w * (2 * c * e + 18 * e), f"{h}x{w} C{c} E{e}"

    _launch("sir_bwd_big_kernel" if h * w > 256 else "sir_bwd_kernel", cost, lambda: hip.call("ddpm_sir_backward", C.byref(d), hip.stream()))
    return dx, d_o, d_e


def _tok_ld(t: Tensor) -> int:
    """row stride (elements) of a token tensor [N, S, ..., width] whose last dim is contiguous (a channel slice of a packed buffer)"""
    assert t.dtype == BF16 and t.stride(-1) == 1
    return t.stride(1)


def attention_forward(q: Tensor, k: Tensor, v: Tensor, heads: int, *, drop_p: float = 0.0, seed: Optional[Tensor] = None,
                      stream_id: int = 0, want_lse: bool = True):
    """q: bf16 [N,S,1,C] (or a channel slice of a packed projection output), k, v: [N,Sk,1,C].  Returns (o [N,S,1,C], lse fp32
    [N,heads,S] or None)."""
    n, s, c = q.shape[0], q.shape[1], q.shape[-1]
    sk = k.shape[1]
    dh = c // heads
    assert dh * heads == c and k.shape[-1] == c and v.shape[-1] == c and v.shape[1] == sk
    o = torch.empty(n, s, 1, c, device=q.device, dtype=BF16)
    lse = torch.empty(n, heads, s, device=q.device, dtype=torch.float32) if want_lse else None
    d = hip.AttnDesc(q=hip.ptr(q), k=hip.ptr(k), v=hip.ptr(v), ldq=_tok_ld(q), ldk=_tok_ld(k), ldv=_tok_ld(v), o=hip.ptr(o), ldo=c,
                     lse=hip.ptr(lse), N=n, S=s, Sk=sk, heads=heads, dh=dh, drop_p=drop_p, seed=hip.ptr(seed), stream_id=stream_id)
    _launch("attn_fwd_kernel", lambda: (2 * (2 * n * s * c + 2 * n * sk * c) + _nb(lse), 4.0 * n * s * sk * c, f"S{s} Sk{sk} h{heads} d{dh}"),
            lambda: hip.call("ddpm_attention_forward", C.byref(d), hip.stream()))
    return o, lse


def attention_backward(q: Tensor, k: Tensor, v: Tensor, o: Tensor, lse: Tensor, dout: Tensor, dq: Tensor, dk: Tensor, dv: Tensor,
                       heads: int, *, drop_p: float = 0.0, seed: Optional[Tensor] = None, stream_id: int = 0) -> None:
    """Gradients of attention_forward into dq / dk / dv (token tensors or channel slices of one packed buffer)."""
    n, s, c = q.shape[0], q.shape[1], q.shape[-1]
    sk = k.shape[1]
    d = hip.AttnDesc(q=hip.ptr(q), k=hip.ptr(k), v=hip.ptr(v), ldq=_tok_ld(q), ldk=_tok_ld(k), ldv=_tok_ld(v), o=hip.ptr(o), ldo=_tok_ld(o),
                     lse=hip.ptr(lse), dout=hip.ptr(dout), ldd=_tok_ld(dout), dq=hip.ptr(dq), dk=hip.ptr(dk), dv=hip.ptr(dv),
                     lddq=_tok_ld(dq), lddk=_tok_ld(dk), lddv=_tok_ld(dv), N=n, S=s, Sk=sk, heads=heads, dh=c // heads, drop_p=drop_p,
                     seed=hip.ptr(seed), stream_id=stream_id)
    _launch("attn_bwd_kernel", lambda: (2 * (4 * n * s * c + 4 * n * sk * c) + _nb(lse), 10.0 * n * s * sk * c, f"S{s} Sk{sk} h{heads} d{c // heads}"),
            lambda: hip.call("ddpm_attention_backward", C.byref(d), hip.stream()))


def sample_norm_forward(a: Tensor, b: Optional[Tensor], act: int, gamma: Tensor, beta: Tensor, eps: float = 1e-5,
                        act_beta: Optional[Tensor] = None):
    """y = GroupNorm(1, C)(act(a) + b) over each sample of bf16 [N,S,1,C].  Returns (y, stats fp32 [N,2])."""
    n, s, c = a.shape[0], a.numel() // (a.shape[0] * a.shape[-1]), a.shape[-1]
    assert a.dtype == BF16 and a.is_contiguous() and (b is None or (b.shape == a.shape and b.is_contiguous()))
    y = torch.empty_like(a)
    stats = torch.empty(n, 2, device=a.device, dtype=torch.float32)
    d = hip.SampleNormDesc(a=hip.ptr(a), b=hip.ptr(b), gamma=hip.ptr(gamma), beta=hip.ptr(beta), y=hip.ptr(y), stats=hip.ptr(stats),
                           N=n, S=s, C=c, act=act, eps=eps, act_beta=_scalar_ptr(act, act_beta))
    _launch("sample_norm_fwd_kernel", lambda: (_nb(a, b, y, gamma, beta, stats), 0.0, f"S{s} C{c} a{act}"),
            lambda: hip.call("ddpm_sample_norm_forward", C.byref(d), hip.stream()))
    return y, stats


def sample_norm_backward(a: Tensor, b: Optional[Tensor], dy: Tensor, act: int, gamma: Tensor, stats: Tensor, part: Tensor, eps: float = 1e-5,
                         act_beta: Optional[Tensor] = None, dbeta_part: Optional[Tensor] = None):
    """Returns (du, da): gradients w.r.t. (act(a) + b) and w.r.t. a (the same tensor when act is the identity).
    part: fp32 [N][3][C] receives d gamma, d beta and the column sums of da per sample."""
    n, s, c = a.shape[0], a.numel() // (a.shape[0] * a.shape[-1]), a.shape[-1]
    assert dy.shape == a.shape and dy.dtype == BF16 and dy.is_contiguous() and part.numel() == n * 3 * c
    du = torch.empty_like(a)
    da = torch.empty_like(a) if act else None
    d = hip.SampleNormDesc(a=hip.ptr(a), b=hip.ptr(b), gamma=hip.ptr(gamma), beta=None, y=None, stats=hip.ptr(stats), dy=hip.ptr(dy),
                           du=hip.ptr(du), da=hip.ptr(da), part=hip.ptr(part), N=n, S=s, C=c, act=act, eps=eps,
                           act_beta=_scalar_ptr(act, act_beta), dbeta_part=hip.ptr(dbeta_part))
    assert act not in hip.ACT_WITH_SCALAR or (dbeta_part is not None and dbeta_part.numel() == n)
    _launch("sample_norm_bwd_kernel", lambda: (_nb(a, b, dy, du, da, gamma, stats, part), 0.0, f"S{s} C{c} a{act}"),
            lambda: hip.call("ddpm_sample_norm_backward", C.byref(d), hip.stream()))
    return du, (da if act else du)


def dwconv3x3(x: Tensor, w9c: Tensor, bias: Optional[Tensor], out: Optional[Tensor] = None, gn_part_groups: int = 0):
    """x: NHWC bf16; w9c: fp32 [9][C] tap-major.  gn_part_groups > 0: returns (y, part) with part = fp32 [N][groups][parts][2]
    partial sums (x, x^2) of y for gn_act_forward(stat_part=part), or None where the launch cannot deliver them."""
    n, h, w, c = x.shape
    assert x.dtype == BF16 and x.is_contiguous() and w9c.shape == (9, c) and w9c.dtype == torch.float32
    if out is None:
        out = torch.empty_like(x)
    parts = hip.lib().ddpm_dwconv3x3_stat_parts(n, h, w, c, gn_part_groups) if gn_part_groups > 0 else 0
    cost = lambda: (_nb(x, w9c, bias, out), 18.0 * x.numel(), f"{h}x{w} C{c}")  # noqa: E731
    if parts > 0:
        part = torch.empty(n, gn_part_groups, parts, 2, device=x.device, dtype=torch.float32)
        _launch("dwconv3x3_kernel", cost, lambda: hip.call("ddpm_dwconv3x3_forward_stats", hip.ptr(x), hip.ptr(w9c), hip.ptr(bias), hip.ptr(out),
                                                           n, h, w, c, gn_part_groups, hip.ptr(part), hip.stream()))
        return out, part
    _launch("dwconv3x3_kernel", cost, lambda: hip.call("ddpm_dwconv3x3_forward", hip.ptr(x), hip.ptr(w9c), hip.ptr(bias), hip.ptr(out), n, h, w, c, hip.stream()))
    return (out, None) if gn_part_groups > 0 else out


def dwconv3x3_affine_supported(x: Tensor, groups: int) -> bool:
    n, h, w, c = x.shape
    return bool(hip.lib().ddpm_dwconv3x3_affine_supported(n, h, w, c, groups))


def dwconv3x3_affine(x: Tensor, affine: Tensor, act: int, w9c: Tensor, bias: Optional[Tensor], groups: int):
    """Depthwise 3x3 of act(x * scale + shift): x = the RAW NHWC bf16 input of a GroupNorm + activation that has not been applied,
    affine = fp32 [N][C][2] from gn_fold_affine (inference on large images; dwconv3x3_affine_supported says where).  Returns
    (y, part): the raw result and its GroupNorm partial sums fp32 [N][groups][parts][2]."""
    n, h, w, c = x.shape
    assert x.dtype == BF16 and x.is_contiguous() and w9c.shape == (9, c) and affine.shape == (n, c, 2) and affine.dtype == torch.float32
    parts = hip.lib().ddpm_dwconv3x3_stat_parts(n, h, w, c, groups)
    out = torch.empty_like(x)
    part = torch.empty(n, groups, parts, 2, device=x.device, dtype=torch.float32)
    _launch("dwconv3x3_kernel", lambda: (_nb(x, w9c, bias, out, affine), 18.0 * x.numel(), f"{h}x{w} C{c} normalise-on-load"),
            lambda: hip.call("ddpm_dwconv3x3_forward_affine", hip.ptr(x), hip.ptr(affine), act, hip.ptr(w9c), hip.ptr(bias), hip.ptr(out),
                             n, h, w, c, groups, hip.ptr(part), hip.stream()))
    return out, part


def dwconv3x3_wgrad_blocks(n: int, h: int, w: int, c: int) -> int:
    return hip.lib().ddpm_dwconv3x3_wgrad_blocks(n, h, w, c)


def dwconv3x3_wgrad(x: Tensor, dy: Tensor, part: Tensor) -> None:
    """part: fp32 [blocks][10][C] per-workgroup partial sums (taps 0..8 of dw, then db)."""
    n, h, w, c = x.shape
    assert dy.shape == x.shape and part.numel() == dwconv3x3_wgrad_blocks(n, h, w, c) * 10 * c
    _launch("dwconv3x3_wgrad_kernel", lambda: (_nb(x, dy, part), 20.0 * x.numel(), f"{h}x{w} C{c}"),
            lambda: hip.call("ddpm_dwconv3x3_wgrad", hip.ptr(x), hip.ptr(dy), hip.ptr(part), n, h, w, c, hip.stream()))


def reduce_rows(src: Tensor, dst: Tensor, entries: Tensor, n_entries: int, max_width: int) -> None:
    """dst[e.dst_off + j] += sum_r src[e.src_off + r*e.row_stride + j] for every entry (device table of ReduceEntry)."""
    hip.call("ddpm_reduce_rows", hip.ptr(src), hip.ptr(dst), hip.ptr(entries), n_entries, max_width, hip.stream())


def reduce_table(entries, device) -> Tensor:
    """Device copy of a list of (src_off, dst_off, rows, row_stride, width) tuples.  row_stride 0 marks a scalar fold: the
    `rows` contiguous floats at src_off are summed into the single number at dst_off."""
    import numpy as np
    arr = np.zeros((len(entries), 4), dtype=np.int64)
    for i, (so, do, rows, stride, width) in enumerate(entries):
        arr[i, 0], arr[i, 1] = so, do
        arr[i, 2] = (rows & 0xFFFFFFFF) | (stride << 32)
        arr[i, 3] = width | ((1 << 32) if stride == 0 else 0)
    return torch.from_numpy(arr).to(device)


def colsum(x: Tensor, out: Tensor, c: int) -> None:
    """out[c] += sum over rows of NHWC bf16 x (any leading dims; x may be a channel slice of a wider packed tensor)."""
    cs = x.stride(-2)
    hip.call("ddpm_colsum", hip.ptr(x), hip.ptr(out), x.numel() // x.shape[-1], c, cs, hip.stream())


def spatial_sum(x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    n, h, w, c = x.shape
    if out is None:
        out = torch.empty(n, c, device=x.device, dtype=torch.float32)
    hip.call("ddpm_spatial_sum", hip.ptr(x), hip.ptr(out), n, h * w, c, hip.stream())
    return out
