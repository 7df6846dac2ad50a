"""GPU parity of GroupNorm(+act/mask/residual) fwd/bwd, depthwise 3x3 fwd/dgrad/wgrad, dense weight-gradient
and the small reductions, each against torch fp32 on CPU with identical bf16-rounded operands."""
import pytest
import torch
import torch.nn.functional as F

from dm_amd.runtime import hip, ops, pack
from oracle import ref_cpu as O
from oracle.noise import hash_normal, hash_uniform
from test_gpu_conv import assert_rel, nchw, nhwc, r16, torch_ref

pytestmark = pytest.mark.gpu


def bits(k):
    return (k.reshape(-1, 8).to(torch.uint8) << torch.arange(8, dtype=torch.uint8)).sum(1).to(torch.uint8)


@pytest.mark.parametrize("n,c,h,g,act", [(2, 32, 16, 16, "gelu"), (3, 64, 8, 16, "silu"), (2, 128, 4, 16, "gelu"),
                                          (5, 512, 2, 32, "gelu"), (2, 256, 2, 16, "none"), (2, 64, 32, 16, "gelu"),
                                          (2, 64, 8, 16, "selu"), (2, 32, 8, 16, "leaky_relu"), (2, 32, 8, 16, "relu6"), (2, 64, 4, 16, "hardsigmoid"),
                                          (2, 64, 4, 16, "relu"), (2, 64, 4, 16, "tanh"), (2, 64, 4, 16, "sigmoid"),
                                          # large images: the many-workgroup (split) path, ragged split sizes, activation-only
                                          (1, 32, 64, 16, "gelu"), (2, 64, 48, 16, "silu"), (2, 64, 40, 0, "gelu"), (1, 128, 128, 16, "gelu"),
                                          (2, 32, 8, 0, "gelu")])
def test_groupnorm_act_fwd_bwd(dev, n, c, h, g, act):
    x = r16(hash_normal((n, c, h, h), 1) * 1.5 + 0.7)
    gamma, beta = hash_normal((c,), 2) * 0.3 + 1, hash_normal((c,), 3) * 0.3
    res = r16(hash_normal((n, c, h, h), 4))
    keep = hash_uniform((n, h, h, c), 5) < 0.9
    gy = r16(hash_normal((n, c, h, h), 6))
    a = O.act_fn(act) if act != "none" else (lambda v: v)

    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z = F.group_norm(xr, g, gr, br, eps=1e-5) if g else xr
    pre = a(z) * keep.permute(0, 3, 1, 2) / 0.9
    ref = pre + res
    grads = torch.autograd.grad(pre, [xr, gr, br] if g else [xr], gy)

    y, stats = ops.gn_act_forward(nhwc(x, dev), gamma.to(dev) if g else None, beta.to(dev) if g else None, g, hip.ACT[act],
                                  res=nhwc(res, dev), mask=bits(keep).to(dev), mask_scale=1 / 0.9)
    assert_rel(nchw(y), ref)
    if g:
        xg = x.view(n, g, -1)
        torch.testing.assert_close(stats[..., 0].cpu(), xg.mean(-1), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(stats[..., 1].cpu(), (xg.var(-1, unbiased=False) + 1e-5).rsqrt(), rtol=1e-3, atol=1e-4)
    part = torch.full((n, 3, c), float("nan"), device=dev)
    dx = ops.gn_act_backward(nhwc(x, dev), nhwc(gy, dev), gamma.to(dev) if g else None, beta.to(dev) if g else None, g, hip.ACT[act],
                             stats, part, mask=bits(keep).to(dev), mask_scale=1 / 0.9)
    assert_rel(nchw(dx), grads[0], 1.5e-2)
    # second stage: sum the per-image partials with the table-driven reduction
    out = torch.zeros(3 * c, device=dev)
    tab = ops.reduce_table([(0, 0, n, 3 * c, c), (c, c, n, 3 * c, c), (2 * c, 2 * c, n, 3 * c, c)], dev)
    ops.reduce_rows(part, out, tab, 3, c)
    dgam, dbet, dbias = out[:c], out[c:2 * c], out[2 * c:]
    assert_rel(dbias.cpu(), grads[0].sum((0, 2, 3)), 2e-2)
    if g:
        assert_rel(dgam.cpu(), grads[1], 1e-2)
        assert_rel(dbet.cpu(), grads[2], 1e-2)


@pytest.mark.parametrize("n,c,h,w_", [(2, 64, 16, 16), (3, 128, 8, 8), (5, 512, 2, 2), (1, 64, 5, 7), (2, 256, 4, 4)])
def test_depthwise_fwd_dgrad_wgrad(dev, n, c, h, w_):
    x = r16(hash_normal((n, c, h, w_), 1))
    w = hash_normal((c, 1, 3, 3), 2) * 0.3
    b = hash_normal((c,), 3) * 0.1
    gy = r16(hash_normal((n, c, h, w_), 4))
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, br, padding=1, groups=c)
    gx, gw, gb = torch.autograd.grad(ref, [xr, wr, br], gy)
    w9 = w.view(c, 9).t().contiguous().to(dev)
    y = ops.dwconv3x3(nhwc(x, dev), w9, b.to(dev))
    assert_rel(nchw(y), ref.detach())
    w9f = w.flip(2, 3).reshape(c, 9).t().contiguous().to(dev)
    dx = ops.dwconv3x3(nhwc(gy, dev), w9f, None)
    assert_rel(nchw(dx), gx)
    nb = ops.dwconv3x3_wgrad_blocks(n, h, w_, c)
    part = torch.full((nb, 10, c), float("nan"), device=dev)
    ops.dwconv3x3_wgrad(nhwc(x, dev), nhwc(gy, dev), part)
    tot = part.sum(0)
    assert_rel(tot[:9].t().reshape(c, 1, 3, 3).cpu(), gw, 5e-3)
    assert_rel(tot[9].cpu(), gb, 5e-3)


@pytest.mark.parametrize("n,c,h,w_,groups,act", [(2, 64, 128, 128, 16, "gelu"), (3, 128, 64, 64, 16, "silu"), (1, 32, 64, 96, 16, "gelu"),
                                                 (2, 256, 64, 64, 4, "relu"), (1, 64, 64, 66, 16, "gelu"), (3, 256, 32, 32, 16, "gelu")])
def test_depthwise_leaves_groupnorm_sums_for_large_images(dev, n, c, h, w_, groups, act):
    """Large images: the depthwise launch stores partial sums of its result per normalisation group, and the GroupNorm that
    follows skips its statistics pass.  Result bit-equal to the plain launch; statistics equal up to fp32 summation order
    (1e-5), the normalised tensor within one bf16 step of the two-pass result and within 1e-2 of torch's F.group_norm."""
    x = r16(hash_normal((n, c, h, w_), 11))
    w = hash_normal((c, 1, 3, 3), 12) * 0.3
    b = hash_normal((c,), 13) * 0.2
    gamma, beta = (1 + 0.3 * hash_normal((c,), 14)).to(dev), (0.3 * hash_normal((c,), 15)).to(dev)
    w9 = w.view(c, 9).t().contiguous().to(dev)
    xd = nhwc(x, dev)
    y, part = ops.dwconv3x3(xd, w9, b.to(dev), gn_part_groups=groups)
    y2 = ops.dwconv3x3(xd, w9, b.to(dev))
    assert part is not None and part.shape[:2] == (n, groups) and torch.equal(y, y2)
    g = y.float().view(n, -1, groups, c // groups)
    torch.testing.assert_close(part[..., 0].sum(2), g.sum((1, 3)), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(part[..., 1].sum(2), (g * g).sum((1, 3)), rtol=1e-4, atol=1e-2)
    z, stats = ops.gn_act_forward(y, gamma, beta, groups, hip.ACT[act], stat_part=part)
    z2, stats2 = ops.gn_act_forward(y2, gamma, beta, groups, hip.ACT[act])
    torch.testing.assert_close(stats, stats2, rtol=1e-5, atol=1e-6)
    assert_rel(z.float().cpu(), z2.float().cpu(), 8e-3)
    assert_rel(nchw(z), O.act_fn(act)(F.group_norm(nchw(y), groups, gamma.cpu(), beta.cpu(), eps=1e-5)), 1e-2)
    # declined: small images, rounds of work items that would straddle images, one-channel groups
    for shape, gr in (((2, 16, 16, 64), 16), ((1, 66, 66, 8), 4), ((1, 64, 64, 32), 32)):
        _, none = ops.dwconv3x3(torch.zeros(shape, device=dev, dtype=torch.bfloat16), torch.zeros(9, shape[3], device=dev), None, gn_part_groups=gr)
        assert none is None


@pytest.mark.parametrize("n,c,h,w_,groups,act", [(3, 64, 64, 64, 16, "gelu"), (2, 64, 128, 128, 16, "silu"), (2, 128, 64, 64, 16, "gelu"),
                                                  (1, 64, 32, 64, 16, "none"), (2, 32, 16, 128, 16, "gelu")])
def test_depthwise_normalises_its_input_on_load(dev, n, c, h, w_, groups, act):
    """Inference on large images: the depthwise kernel reads the RAW result of the expansion and applies GroupNorm + activation
    on the way into its LDS row ring.  Against the two-launch path (GroupNorm launch, depthwise kernel with sums): the result
    within one bf16 step of its scale, the GroupNorm sums it leaves equal to the sums of what it stored; against torch fp32
    within the bf16 tolerance."""
    x = r16(hash_normal((n, c, h, w_), 301) * 1.2 + 0.3)
    gamma, beta = (1 + 0.3 * hash_normal((c,), 302)), (0.3 * hash_normal((c,), 303))
    wt = hash_normal((c, 1, 3, 3), 304) / 3
    bias = hash_normal((c,), 305) * 0.1
    xd = nhwc(x, dev)
    assert ops.dwconv3x3_affine_supported(xd, groups)
    parts = 8
    g = xd.float().view(n, parts, h * w_ // parts, groups, c // groups)
    part = torch.stack([g.sum((2, 4)), (g * g).sum((2, 4))], -1).permute(0, 2, 1, 3).contiguous()
    _, affine = ops.gn_fold_affine(part, gamma.to(dev), beta.to(dev), h * w_)
    w9 = wt.view(c, 9).t().contiguous().to(dev)
    y_gn, _ = ops.gn_act_forward(xd, gamma.to(dev), beta.to(dev), groups, hip.ACT[act])
    ref_y, ref_part = ops.dwconv3x3(y_gn, w9, bias.to(dev), gn_part_groups=groups)
    got, gpart = ops.dwconv3x3_affine(xd, affine, hip.ACT[act], w9, bias.to(dev), groups)
    assert_rel(got.float().cpu(), ref_y.float().cpu(), 8e-3)
    assert gpart.shape == ref_part.shape
    gg = got.float().view(n, -1, groups, c // groups)
    torch.testing.assert_close(gpart[..., 0].sum(2), gg.sum((1, 3)), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(gpart[..., 1].sum(2), (gg * gg).sum((1, 3)), rtol=1e-4, atol=1e-2)
    a = (lambda v: v) if act == "none" else O.act_fn(act)
    ref = F.conv2d(r16(a(F.group_norm(x, groups, gamma, beta, eps=1e-5))), wt, bias, padding=1, groups=c)
    assert_rel(nchw(got), ref, 1e-2)
    assert not ops.dwconv3x3_affine_supported(nhwc(r16(hash_normal((1, 64, 8, 8), 1)), dev), groups)


WG_CASES = [
    (hip.CONV_3X3_S1, 2, 32, 32, 16, 16),
    (hip.CONV_3X3_S1, 3, 64, 32, 8, 8),
    (hip.CONV_3X3_S1, 5, 256, 512, 2, 2),
    (hip.CONV_3X3_S1, 2, 16, 32, 32, 32),
    (hip.CONV_3X3_S1, 2, 32, 3, 32, 32),
    (hip.CONV_3X3_S1, 1, 32, 64, 12, 20),
    (hip.CONV_3X3_S2, 2, 32, 64, 16, 16),
    (hip.CONV_3X3_S2, 7, 128, 256, 4, 4),
    (hip.CONV_1X1, 2, 32, 64, 16, 16),
    (hip.CONV_1X1, 2, 512, 256, 2, 2),
    (hip.CONV_1X1, 9, 32, 128, 1, 1),
    (hip.CONV_1X1, 300, 64, 32, 8, 8),
    (hip.CONVT_2X2_S2, 2, 64, 32, 8, 8),
    (hip.CONVT_2X2_S2, 3, 256, 128, 2, 2),
]


def unpack_dw(kind, dwk, co, ci):
    """kernel-layout fp32 [slices][co][ci_padded] -> torch layout"""
    dwk = dwk[:, :, :ci]
    if kind in (hip.CONV_3X3_S1, hip.CONV_3X3_S2):
        return dwk.view(3, 3, co, ci).permute(2, 3, 0, 1)
    if kind == hip.CONV_1X1:
        return dwk.view(co, ci, 1, 1)
    return dwk.view(2, 2, co, ci).permute(3, 2, 0, 1)


@pytest.mark.parametrize("kind,n,ci,co,h,w_", WG_CASES)
def test_conv_wgrad(dev, kind, n, ci, co, h, w_):
    k = {hip.CONV_3X3_S1: 3, hip.CONV_3X3_S2: 3, hip.CONV_1X1: 1, hip.CONVT_2X2_S2: 2}[kind]
    wshape = (ci, co, k, k) if kind == hip.CONVT_2X2_S2 else (co, ci, k, k)
    w = torch.zeros(wshape, requires_grad=True)
    x = r16(hash_normal((n, ci, h, w_), 3))
    ref = torch_ref(kind, x, w, None)
    gy = r16(hash_normal(tuple(ref.shape), 4))
    (gw,) = torch.autograd.grad(ref, w, gy)
    slices = {3: 9, 1: 1, 2: 4}[k]
    dwk = torch.zeros(slices, co, ci, device=dev)
    ops.conv_wgrad(kind, nhwc(x, dev), nhwc(gy, dev, (co + 7) // 8 * 8), co, dwk)
    assert_rel(unpack_dw(kind, dwk.cpu(), co, ci), gw, 5e-3)
    # accumulation semantics: a second call adds
    ops.conv_wgrad(kind, nhwc(x, dev), nhwc(gy, dev, (co + 7) // 8 * 8), co, dwk)
    assert_rel(unpack_dw(kind, dwk.cpu(), co, ci), 2 * gw, 5e-3)


@pytest.mark.parametrize("n,ci,co,w_", [(16, 64, 128, 8), (32, 128, 128, 4), (9, 64, 256, 16), (40, 128, 256, 8), (64, 192, 128, 4), (8, 64, 128, 16),
                                         (160, 256, 256, 8), (24, 128, 256, 16), (512, 256, 256, 4)])
def test_conv_wgrad_pipelined_kernel(dev, n, ci, co, w_):
    """The pipelined weight-gradient kernel (128 x 64 channel tiles, all nine taps in accumulators, two LDS stages), forced
    onto small problems: atomics into dw, and per-split partial blocks folded on the host, against torch fp32 on the same
    bf16 operands and against the tap-list kernel."""
    w = torch.zeros(co, ci, 3, 3, requires_grad=True)
    x = r16(hash_normal((n, ci, w_, w_), 3))
    ref = torch_ref(hip.CONV_3X3_S1, x, w, None)
    gy = r16(hash_normal(tuple(ref.shape), 4))
    (gw,) = torch.autograd.grad(ref, w, gy)
    xd, gd = nhwc(x, dev), nhwc(gy, dev)
    old = torch.zeros(9, co, ci, device=dev)
    prev = hip.lib().ddpm_conv_pipe_mode(-1)
    try:
        ops.conv_wgrad(hip.CONV_3X3_S1, xd, gd, co, old)
        hip.lib().ddpm_conv_pipe_mode(1)
        splits = ops.conv_wgrad_splits(hip.CONV_3X3_S1, xd, gd, co)
        assert splits % 8 == 0 and splits >= 8, splits               # (the tap-list kernel's count is not a multiple of 8 for these sizes)
        dwk = torch.zeros(9, co, ci, device=dev)
        ops.conv_wgrad(hip.CONV_3X3_S1, xd, gd, co, dwk)
        part = torch.full((splits, 9, co, ci), float("nan"), device=dev)
        ops.conv_wgrad(hip.CONV_3X3_S1, xd, gd, co, None, part=part)
    finally:
        hip.lib().ddpm_conv_pipe_mode(prev)
    assert_rel(unpack_dw(hip.CONV_3X3_S1, dwk.cpu(), co, ci), gw, 5e-3)
    assert_rel(unpack_dw(hip.CONV_3X3_S1, part.sum(0).cpu(), co, ci), gw, 5e-3)
    assert_rel(dwk.cpu(), old.cpu(), 1e-4)                           # same bf16 products, fp32 sums in another order


@pytest.mark.parametrize("n,ci,co,h,w_,with_add", [(3, 16, 32, 32, 32, False), (2, 32, 32, 64, 64, True), (5, 32, 3, 32, 32, False),
                                                    (1, 32, 32, 128, 128, False), (70, 32, 32, 32, 32, True), (2, 16, 32, 16, 64, False),
                                                    (1, 32, 1, 8, 128, False)])
def test_conv_wgrad_image_end_kernel(dev, n, ci, co, h, w_, with_add):
    """The weight-gradient kernel of the thin 3x3 convolutions at the image ends (stem: 16 stored input channels; first-stage and
    feature convolutions 32 -> 32, with the time-embedding add of the forward loader; head: 3 or 1 of 16 stored output channels;
    unets.py:78-85,160-167 of the reference), forced onto small problems: atomics into dw and per-workgroup partial blocks,
    against torch fp32 on the same bf16 operands and against the tap-list kernel.  (70 images: more bands than workgroups.)"""
    w = torch.zeros(co, ci, 3, 3, requires_grad=True)
    x = r16(hash_normal((n, ci, h, w_), 3))
    temb = hash_normal((n, ci), 5) if with_add else None
    xin = x if temb is None else r16(x + temb[:, :, None, None])
    ref = F.conv2d(xin, w, None, padding=1)
    gy = r16(hash_normal(tuple(ref.shape), 4))
    (gw,) = torch.autograd.grad(ref, w, gy)
    xd, gd = nhwc(x, dev), nhwc(gy, dev, (co + 15) // 16 * 16)
    kw = {} if temb is None else {"in_add": temb.to(dev)}
    old = torch.zeros(9, co, ci, device=dev)
    prev = hip.lib().ddpm_conv_pipe_mode(-1)
    try:
        assert ops.conv_wgrad_pipe_splits(hip.CONV_3X3_S1, xd, gd, co, **kw) == 0
        ops.conv_wgrad(hip.CONV_3X3_S1, xd, gd, co, old, **kw)
        hip.lib().ddpm_conv_pipe_mode(1)
        splits = ops.conv_wgrad_pipe_splits(hip.CONV_3X3_S1, xd, gd, co, **kw)
        assert splits == min(256, n * h * w_ // 256), splits
        assert ops.conv_wgrad_splits(hip.CONV_3X3_S1, xd, gd, co) == splits
        dwk = torch.zeros(9, co, ci, device=dev)
        ops.conv_wgrad(hip.CONV_3X3_S1, xd, gd, co, dwk, **kw)
        part = torch.full((splits, 9, co, ci), float("nan"), device=dev)
        ops.conv_wgrad(hip.CONV_3X3_S1, xd, gd, co, None, part=part, **kw)
        hip.lib().ddpm_conv_pipe_mode(0)
        small = n * h * w_ < 65536
        assert (ops.conv_wgrad_pipe_splits(hip.CONV_3X3_S1, xd, gd, co, **kw) == 0) == small       # by size: one band per CU and more
    finally:
        hip.lib().ddpm_conv_pipe_mode(prev)
    assert_rel(unpack_dw(hip.CONV_3X3_S1, dwk.cpu(), co, ci), gw, 5e-3)
    assert_rel(unpack_dw(hip.CONV_3X3_S1, part.sum(0).cpu(), co, ci), gw, 5e-3)
    assert_rel(dwk.cpu(), old.cpu(), 1e-4)                           # same bf16 products, fp32 sums in another order


def test_conv_wgrad_pipelined_kernel_two_sources(dev):
    """concatenated input (two NHWC sources): an input-channel block of the pipelined kernel reads from one of them"""
    n, c0, c1, co, h = 24, 64, 128, 128, 8
    x0, x1 = r16(hash_normal((n, c0, h, h), 1)), r16(hash_normal((n, c1, h, h), 2))
    w = torch.zeros(co, c0 + c1, 3, 3, requires_grad=True)
    ref = F.conv2d(torch.cat([x0, x1], 1), w, None, padding=1)
    gy = r16(hash_normal(tuple(ref.shape), 4))
    (gw,) = torch.autograd.grad(ref, w, gy)
    dwk = torch.zeros(9, co, c0 + c1, device=dev)
    prev = hip.lib().ddpm_conv_pipe_mode(1)
    try:
        assert ops.conv_wgrad_pipe_splits(hip.CONV_3X3_S1, nhwc(x0, dev), nhwc(gy, dev), co, x1=nhwc(x1, dev)) > 0
        ops.conv_wgrad(hip.CONV_3X3_S1, nhwc(x0, dev), nhwc(gy, dev), co, dwk, x1=nhwc(x1, dev))
    finally:
        hip.lib().ddpm_conv_pipe_mode(prev)
    assert_rel(unpack_dw(hip.CONV_3X3_S1, dwk.cpu(), co, c0 + c1), gw, 5e-3)


@pytest.mark.parametrize("n,h", [(24, 8), (64, 4), (10, 16)])
def test_conv_wgrad_pipelined_kernel_fused_input(dev, n, h):
    """the forward loader's transforms of the first source (time-embedding add, then dropout mask) inside the pipelined kernel"""
    c0, c1, co = 64, 64, 128
    x0, x1 = r16(hash_normal((n, c0, h, h), 1)), r16(hash_normal((n, c1, h, h), 2))
    temb = hash_normal((n, c0), 3)
    keep = hash_uniform((n, h, h, c0), 6) < 0.9
    xin = torch.cat([r16(r16(x0 + temb[:, :, None, None]) * keep.permute(0, 3, 1, 2) / 0.9), x1], 1)
    w = torch.zeros(co, c0 + c1, 3, 3, requires_grad=True)
    ref = F.conv2d(xin, w, None, padding=1)
    gy = r16(hash_normal(tuple(ref.shape), 4))
    (gw,) = torch.autograd.grad(ref, w, gy)
    dwk = torch.zeros(9, co, c0 + c1, device=dev)
    prev = hip.lib().ddpm_conv_pipe_mode(1)
    try:
        assert ops.conv_wgrad_pipe_splits(hip.CONV_3X3_S1, nhwc(x0, dev), nhwc(gy, dev), co, x1=nhwc(x1, dev), in_add=temb.to(dev)) > 0
        ops.conv_wgrad(hip.CONV_3X3_S1, nhwc(x0, dev), nhwc(gy, dev), co, dwk, x1=nhwc(x1, dev), in_add=temb.to(dev),
                       in_mask=bits(keep).to(dev), in_scale=1 / 0.9)
    finally:
        hip.lib().ddpm_conv_pipe_mode(prev)
    assert_rel(unpack_dw(hip.CONV_3X3_S1, dwk.cpu(), co, c0 + c1), gw, 5e-3)


def test_conv_wgrad_fused_input(dev):
    n, c0, c1, co, h = 3, 64, 64, 64, 8
    x0, x1 = r16(hash_normal((n, c0, h, h), 1)), r16(hash_normal((n, c1, h, h), 2))
    temb = hash_normal((n, c0), 3)
    keep = hash_uniform((n, h, h, c0), 6) < 0.9
    xin = torch.cat([r16(r16(x0 + temb[:, :, None, None]) * keep.permute(0, 3, 1, 2) / 0.9), x1], 1)
    w = torch.zeros(co, c0 + c1, 3, 3, requires_grad=True)
    ref = F.conv2d(xin, w, None, padding=1)
    gy = r16(hash_normal(tuple(ref.shape), 4))
    (gw,) = torch.autograd.grad(ref, w, gy)
    dwk = torch.zeros(9, co, c0 + c1, device=dev)
    ops.conv_wgrad(hip.CONV_3X3_S1, nhwc(x0, dev), nhwc(gy, dev), co, dwk, x1=nhwc(x1, dev), in_add=temb.to(dev),
                   in_mask=bits(keep).to(dev), in_scale=1 / 0.9)
    assert_rel(unpack_dw(hip.CONV_3X3_S1, dwk.cpu(), co, c0 + c1), gw, 5e-3)


def test_small_reductions_and_masks(dev):
    x = r16(hash_normal((5, 7, 3, 32), 1))          # NHWC directly
    out = torch.zeros(20, device=dev)
    xd = x.to(dev).bfloat16()
    ops.colsum(xd, out, 20)
    assert_rel(out.cpu(), x.reshape(-1, 32).sum(0)[:20], 5e-3)
    ss = ops.spatial_sum(xd)
    assert_rel(ss.cpu(), x.sum((1, 2)), 5e-3)
    # dropout masks: deterministic in the seed, right keep rate, independent across sites
    sizes = [4096, 8192, 1000 * 4]
    offs = torch.tensor([0, sizes[0], sizes[0] + sizes[1], sum(sizes)], device=dev, dtype=torch.long)
    seed = torch.tensor([77], device=dev, dtype=torch.long)
    m1 = torch.zeros(sum(sizes), device=dev, dtype=torch.uint8)
    m2 = torch.zeros_like(m1)
    hip.call("ddpm_dropout_masks", hip.ptr(m1), hip.ptr(offs), 3, sum(sizes), 0.9, hip.ptr(seed), hip.stream())
    hip.call("ddpm_dropout_masks", hip.ptr(m2), hip.ptr(offs), 3, sum(sizes), 0.9, hip.ptr(seed), hip.stream())
    assert torch.equal(m1, m2)
    bitsum = sum(((m1 >> j) & 1).sum().item() for j in range(8))
    assert abs(bitsum / (8 * sum(sizes)) - 0.9) < 0.01
    assert not torch.equal(m1[:4096], m1[4096:8192])
    hip.call("ddpm_bump_seed", hip.ptr(seed), hip.stream())
    hip.call("ddpm_dropout_masks", hip.ptr(m2), hip.ptr(offs), 3, sum(sizes), 0.9, hip.ptr(seed), hip.stream())
    assert not torch.equal(m1, m2)


@pytest.mark.parametrize("n,h,w,c,act", [(3, 32, 32, 32, "gelu"), (2, 16, 16, 64, "silu"), (2, 6, 10, 16, "relu"), (1, 64, 64, 32, "gelu")])
def test_act_backward_streaming(dev, n, h, w, c, act):
    """dx = dy * act'(x) and its per-range column sums (the bare activation of a conv -> act unit), against torch fp32 and
    against the GroupNorm kernel's groups = 0 path.  Tolerance 1e-2 of the scale (bf16 result), sums 1e-3."""
    from oracle import ref_cpu as O
    from oracle.noise import hash_normal
    x = hash_normal((n, h, w, c), 91).bfloat16()
    dy = hash_normal((n, h, w, c), 92).bfloat16()
    xl = x.float().requires_grad_(True)
    O.act_fn(act)(xl).backward(dy.float())
    xd, dyd = x.to(dev), dy.to(dev)
    p = ops.act_backward_splits(xd)
    assert p >= 1 and (h * w) % p == 0
    part = torch.full((n * p, c), float("nan"), device=dev)
    dx = ops.act_backward(xd, dyd, hip.ACT[act], part, p)
    scale = xl.grad.abs().max().item()
    assert (dx.float().cpu() - xl.grad).abs().max().item() <= 1e-2 * scale
    torch.testing.assert_close(part.view(n, p, c).sum(1).cpu(), xl.grad.sum((1, 2)), rtol=1e-3, atol=1e-3 * h * w ** 0.5)
    part3 = torch.zeros(n, 3, c, device=dev)
    dx0 = ops.gn_act_backward(xd, dyd, None, None, 0, hip.ACT[act], None, part3)
    assert torch.equal(dx, dx0)
    assert ops.act_backward_splits(torch.empty(1, 4, 4, 128, device=dev, dtype=torch.bfloat16)) == 0
