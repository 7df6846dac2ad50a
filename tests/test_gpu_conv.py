"""GPU parity of the MFMA implicit-GEMM convolution (forward and data-gradient of every kind) against
torch fp32 CPU convolutions of the same bf16-rounded operands.  Tolerance: 1e-2 of the output scale
(bf16 storage of the result; accumulation is fp32)."""
import pytest
import torch
import torch.nn.functional as F

from dm_amd.runtime import hip, ops, pack
from oracle.noise import hash_normal, hash_uniform

pytestmark = pytest.mark.gpu


def r16(t):
    return t.bfloat16().float()


def nhwc(t, dev, cpad=None):
    t = t.permute(0, 2, 3, 1)
    if cpad and cpad > t.shape[3]:
        t = F.pad(t, (0, cpad - t.shape[3]))
    return t.contiguous().to(dev).bfloat16()


def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2)


def assert_rel(got, ref, tol=1e-2):
    scale = ref.abs().max().item() + 1e-6
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"max err {err:.4g} vs scale {scale:.4g}"


CASES = [
    # kind, N, Cin, Cout, H, W
    (hip.CONV_3X3_S1, 2, 32, 32, 16, 16),
    (hip.CONV_3X3_S1, 3, 64, 32, 8, 8),
    (hip.CONV_3X3_S1, 5, 128, 64, 8, 8),
    (hip.CONV_3X3_S1, 5, 256, 512, 2, 2),
    (hip.CONV_3X3_S1, 2, 16, 32, 32, 32),     # stem-like (3 real channels padded to 16)
    (hip.CONV_3X3_S1, 2, 32, 3, 32, 32),      # image head
    (hip.CONV_3X3_S1, 1, 32, 64, 12, 20),     # ragged tiles
    (hip.CONV_3X3_S2, 2, 32, 64, 16, 16),
    (hip.CONV_3X3_S2, 7, 128, 256, 4, 4),
    (hip.CONV_1X1, 2, 32, 64, 16, 16),
    (hip.CONV_1X1, 2, 512, 256, 2, 2),
    (hip.CONV_1X1, 9, 32, 128, 1, 1),         # nn.Linear of the time-embedding MLP
    (hip.CONVT_2X2_S2, 2, 64, 32, 8, 8),
    (hip.CONVT_2X2_S2, 3, 256, 128, 2, 2),
]


def torch_ref(kind, x, w, b):
    if kind == hip.CONV_3X3_S1:
        return F.conv2d(x, w, b, padding=1)
    if kind == hip.CONV_3X3_S2:
        return F.conv2d(x, w, b, stride=2, padding=1)
    if kind == hip.CONV_1X1:
        return F.conv2d(x, w, b)
    return F.conv_transpose2d(x, w, b, stride=2)


@pytest.mark.parametrize("kind,n,ci,co,h,w_", CASES)
def test_conv_forward_and_dgrad(dev, kind, n, ci, co, h, w_):
    k = {hip.CONV_3X3_S1: 3, hip.CONV_3X3_S2: 3, hip.CONV_1X1: 1, hip.CONVT_2X2_S2: 2}[kind]
    wshape = (ci, co, k, k) if kind == hip.CONVT_2X2_S2 else (co, ci, k, k)
    w = r16(hash_normal(wshape, 1) * (1.0 / (ci * k * k)) ** 0.5)
    if ci == 16:                                # padded input channels carry zero weights
        w[:, 3:] = 0
    b = hash_normal((co,), 2) * 0.1
    x = r16(hash_normal((n, ci, h, w_), 3))
    ref = torch_ref(kind, x, w, b)
    y = ops.conv_forward(kind, nhwc(x, dev), pack.pack_fwd(kind, w.to(dev)), b.to(dev), co, out_cs=(co + 7) // 8 * 8)
    assert_rel(nchw(y)[:, :co], ref)
    # data gradient: dX = conv_dgrad(dY)
    gy = r16(hash_normal(tuple(ref.shape), 4))
    xg = x.clone().requires_grad_(True)
    (gref,) = torch.autograd.grad(torch_ref(kind, xg, w, None), xg, gy)
    copad = (co + 15) // 16 * 16
    gx = ops.conv_forward(pack.DGRAD_KIND[kind], nhwc(gy, dev, copad), pack.pack_dgrad(kind, w.to(dev)), None, ci)
    assert_rel(nchw(gx), gref)


def test_conv_fused_input_and_epilogue(dev):
    """two-source concat + time-embedding add + input dropout mask; activation, output mask, residual."""
    n, c0, c1, co, h = 3, 64, 64, 64, 8
    x0, x1 = r16(hash_normal((n, c0, h, h), 1)), r16(hash_normal((n, c1, h, h), 2))
    temb = hash_normal((n, c0), 3)
    w = r16(hash_normal((co, c0 + c1, 3, 3), 4) * 0.04)
    b = hash_normal((co,), 5) * 0.1
    keep = (hash_uniform((n, h, h, c0), 6) < 0.9)
    res = r16(hash_normal((n, co, h, h), 7))
    okeep = (hash_uniform((n, h, h, co), 8) < 0.8)

    def bits(k):
        return (k.reshape(-1, 8).to(torch.uint8) << torch.arange(8, dtype=torch.uint8)).sum(1).to(torch.uint8)

    xin = r16(x0 + temb[:, :, None, None]) * keep.permute(0, 3, 1, 2) / 0.9
    ref = F.gelu(F.conv2d(torch.cat([r16(xin), x1], 1), w, b, padding=1))
    ref = ref * okeep.permute(0, 3, 1, 2) / 0.8 + res
    y = ops.conv_forward(hip.CONV_3X3_S1, nhwc(x0, dev), pack.pack_fwd(hip.CONV_3X3_S1, w.to(dev)), b.to(dev), co, x1=nhwc(x1, dev),
                         in_add=temb.to(dev), in_mask=bits(keep).to(dev), in_scale=1 / 0.9, act=hip.ACT["gelu"], res=nhwc(res, dev),
                         out_mask=bits(okeep).to(dev), out_scale=1 / 0.8)
    assert_rel(nchw(y), ref, 1.5e-2)


@pytest.mark.parametrize("n,ci,co,h,wd", [(2, 32, 3, 16, 16), (3, 32, 1, 32, 32), (1, 32, 3, 20, 72), (2, 64, 4, 8, 40), (1, 32, 1, 128, 128),
                                            (2, 16, 3, 24, 40), (2, 48, 2, 16, 16), (3, 32, 2, 64, 64)])
def test_conv_head_f32_nchw_tanh(dev, n, ci, co, h, wd):
    """the image head (few output channels, tanh, fp32 NCHW): the matrix-core kernel with zero-padded weight rows (16 / 32 / 64
    input channels) and the vector-ALU kernel (48 here), ragged tiles included"""
    x = r16(hash_normal((n, ci, h, wd), 1))
    w = r16(hash_normal((co, ci, 3, 3), 2) * 0.06)
    b = hash_normal((co,), 3) * 0.1
    ref = torch.tanh(F.conv2d(x, w, b, padding=1))
    y = ops.conv_forward(hip.CONV_3X3_S1, nhwc(x, dev), pack.pack_fwd(hip.CONV_3X3_S1, w.to(dev)), b.to(dev), co, act=hip.ACT["tanh"],
                         f32_nchw=True)
    torch.testing.assert_close(y.cpu(), ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n,h,wd,act,with_res,second", [(2, 16, 16, "none", False, False), (1, 12, 20, "gelu", False, False), (3, 32, 32, "none", True, False),
                                                        (2, 40, 72, "none", False, True), (1, 64, 64, "silu", False, False), (5, 8, 8, "none", False, True)])
def test_conv3x3_32_to_32_kernel(dev, n, h, wd, act, with_res, second):
    """3x3 stride 1, 32 -> 32 channels: the kernel with all nine weight slices in registers (ragged tiles, activation, residual,
    second activated result), against torch fp32 on the same bf16 operands"""
    from oracle import ref_cpu as O
    x = r16(hash_normal((n, 32, h, wd), 11))
    w = r16(hash_normal((32, 32, 3, 3), 12) * (1.0 / 288) ** 0.5)
    b = hash_normal((32,), 13) * 0.2
    res = r16(hash_normal((n, 32, h, wd), 14))
    ref = F.conv2d(x, w, b, padding=1)
    kw = {}
    if act != "none":
        kw["act"] = hip.ACT[act]
        ref = O.act_fn(act)(ref)
    if with_res:
        kw["res"] = nhwc(res, dev)
        ref = r16(ref) + res
    if second:
        kw["act2"] = hip.ACT["gelu"]
    got = ops.conv_forward(hip.CONV_3X3_S1, nhwc(x, dev), pack.pack_fwd(hip.CONV_3X3_S1, w.to(dev)), b.to(dev), 32, **kw)
    assert hip.lib().ddpm_conv_last_kernel() == 3
    if second:
        raw, y = got
        assert_rel(nchw(raw), ref)
        assert_rel(nchw(y), F.gelu(nchw(raw)), 8e-3)
    else:
        assert_rel(nchw(got), ref)


def test_conv_rejects_bad_shapes(dev):
    x = torch.zeros(1, 4, 4, 24, device=dev, dtype=torch.bfloat16)
    w = torch.zeros(1, 32, 24, device=dev, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="multiples of 16"):
        ops.conv_forward(hip.CONV_1X1, x, w, None, 32)


@pytest.mark.parametrize("n,ci,c1,co,hw,groups,act,with_res,with_add", [
    (9, 256, 0, 256, 2, 16, "gelu", False, True),      # ragged last tile (9 images of 4 pixels: 36 rows)
    (5, 128, 0, 128, 4, 16, "silu", True, False),
    (4, 256, 256, 256, 2, 16, "gelu", False, True),    # decoder: concatenated skip connection
    (3, 256, 0, 512, 2, 32, "relu", False, False),
    (5, 64, 0, 64, 8, 16, "gelu", False, True),        # 8x8: one image per 64-row tile
    (3, 64, 64, 64, 8, 16, "silu", True, False),
    (2, 128, 0, 128, 8, 16, "gelu", False, False),
])
def test_conv_with_fused_groupnorm(dev, n, ci, c1, co, hw, groups, act, with_res, with_add):
    """conv -> GroupNorm -> act (+ residual) finished inside the convolution's launch (2x2 / 4x4 images): raw result,
    statistics and normalised output against torch fp32 on the same bf16-rounded operands.  Tolerance 1e-2 of the scale
    for the bf16 tensors, 1e-3 relative for the statistics."""
    from oracle import ref_cpu as O
    x = r16(hash_normal((n, ci, hw, hw), 61))
    xb = r16(hash_normal((n, c1, hw, hw), 62)) if c1 else None
    w = r16(hash_normal((co, ci + c1, 3, 3), 63) * (1.5 / (3 * (ci + c1) ** 0.5)))
    b = hash_normal((co,), 64) * 0.1
    gamma, beta = 1 + 0.3 * hash_normal((co,), 65), 0.3 * hash_normal((co,), 66)
    res = r16(hash_normal((n, co, hw, hw), 67)) if with_res else None
    add = hash_normal((n, ci), 68) * 0.5 if with_add else None
    xin = x if add is None else r16(x + add[:, :, None, None])
    raw_ref = r16(F.conv2d(xin if xb is None else torch.cat([xin, xb], 1), w, b, padding=1))
    y_ref = O.act_fn(act)(F.group_norm(raw_ref, groups, gamma, beta, eps=1e-5))
    if with_res:
        y_ref = y_ref + res
    g = raw_ref.view(n, groups, -1)
    mean_ref, rstd_ref = g.mean(2), (g.var(2, unbiased=False) + 1e-5).rsqrt()

    x0d, x1d = nhwc(x, dev), (nhwc(xb, dev) if xb is not None else None)
    assert ops.conv_gn_fusable(hip.CONV_3X3_S1, x0d, x1d, co, groups)
    raw, y, stats = ops.conv_forward(hip.CONV_3X3_S1, x0d, pack.pack_fwd(hip.CONV_3X3_S1, w).to(dev), b.to(dev), co, x1=x1d,
                                     in_add=None if add is None else add.to(dev),
                                     gn=(gamma.to(dev), beta.to(dev), groups, hip.ACT[act], None if res is None else nhwc(res, dev)))
    assert_rel(nchw(raw), raw_ref)
    assert_rel(nchw(y), y_ref, 1.5e-2)
    torch.testing.assert_close(stats[..., 0].cpu(), mean_ref, rtol=2e-2, atol=2e-3)
    torch.testing.assert_close(stats[..., 1].cpu(), rstd_ref, rtol=1e-2, atol=1e-3)
    # and the two-launch path gives the same tensors
    raw2 = ops.conv_forward(hip.CONV_3X3_S1, x0d, pack.pack_fwd(hip.CONV_3X3_S1, w).to(dev), b.to(dev), co, x1=x1d,
                            in_add=None if add is None else add.to(dev))
    y2, stats2 = ops.gn_act_forward(raw2, gamma.to(dev), beta.to(dev), groups, hip.ACT[act], res=None if res is None else nhwc(res, dev))
    assert torch.equal(raw, raw2)
    torch.testing.assert_close(stats, stats2, rtol=1e-4, atol=1e-5)
    assert_rel(y.float().cpu(), y2.float().cpu(), 1e-2)
    # inference: the raw result is not stored, the normalised one and the statistics are the same bits
    none, y3, stats3 = ops.conv_forward(hip.CONV_3X3_S1, x0d, pack.pack_fwd(hip.CONV_3X3_S1, w).to(dev), b.to(dev), co, x1=x1d,
                                        in_add=None if add is None else add.to(dev), keep_raw=False,
                                        gn=(gamma.to(dev), beta.to(dev), groups, hip.ACT[act], None if res is None else nhwc(res, dev)))
    assert none is None and torch.equal(y3, y) and torch.equal(stats3, stats)
    assert not ops.conv_gn_fusable(hip.CONV_3X3_S1, torch.empty(2, 64, 64, 32, device=dev, dtype=torch.bfloat16), None, 32, 16)   # image larger than a tile


@pytest.mark.parametrize("n,c_used,co,h,wd,act", [(3, 1, 32, 32, 32, "gelu"), (2, 3, 32, 20, 72, "none"), (1, 3, 32, 64, 64, "silu"), (2, 4, 24, 8, 8, "none")])
def test_conv_thin_input(dev, n, c_used, co, h, wd, act):
    """3x3 convolution of a zero-padded image tensor (16 stored channels, c_used real ones) to 32 channels: the stem and
    the head's data-gradient, on the vector-ALU kernel (C0_used promise) - against torch and against the general kernel."""
    from oracle import ref_cpu as O
    x = r16(hash_normal((n, c_used, h, wd), 71))
    w = r16(hash_normal((co, c_used, 3, 3), 72) * 0.3)
    b = hash_normal((co,), 73) * 0.1
    f = O.act_fn(act) if act != "none" else (lambda t: t)
    ref = f(F.conv2d(x, w, b, padding=1))
    w16 = torch.zeros(co, 16, 3, 3); w16[:, :c_used] = w
    xd, wp = nhwc(x, dev, cpad=16), pack.pack_fwd(hip.CONV_3X3_S1, w16).to(dev)
    y = ops.conv_forward(hip.CONV_3X3_S1, xd, wp, b.to(dev), co, act=hip.ACT[act], c0_used=c_used, out_cs=32)
    y0 = ops.conv_forward(hip.CONV_3X3_S1, xd, wp, b.to(dev), co, act=hip.ACT[act], out_cs=32)
    assert_rel(nchw(y[..., :co]), ref)
    assert_rel(y[..., :co].float().cpu(), y0[..., :co].float().cpu(), 1e-2)


def test_conv_second_activated_result(dev):
    """conv -> act units without normalisation: raw and act(raw) from one launch (general kernel and thin-input kernel),
    identical to the raw result of a plain launch and to the activation of its bf16 values."""
    from oracle import ref_cpu as O
    for ci, c_used, h in [(32, 0, 32), (16, 1, 32), (16, 3, 16)]:
        x = r16(hash_normal((2, ci, h, h), 81))
        if c_used:
            x[:, c_used:] = 0
        w = r16(hash_normal((32, ci, 3, 3), 82) * 0.1)
        b = hash_normal((32,), 83) * 0.1
        xd, wp = nhwc(x, dev), pack.pack_fwd(hip.CONV_3X3_S1, w).to(dev)
        raw, y = ops.conv_forward(hip.CONV_3X3_S1, xd, wp, b.to(dev), 32, act2=hip.ACT["gelu"], c0_used=c_used)
        raw0 = ops.conv_forward(hip.CONV_3X3_S1, xd, wp, b.to(dev), 32, c0_used=c_used)
        assert torch.equal(raw, raw0)
        assert_rel(nchw(raw), F.conv2d(x, w, b, padding=1))
        assert_rel(y.float().cpu(), O.act_fn("gelu")(raw.float().cpu()), 1e-2)


@pytest.mark.parametrize("n,c,c0,hw,groups,act", [(9, 256, 256, 2, 16, "gelu"), (5, 128, 128, 4, 16, "silu"), (3, 64, 64, 8, 16, "gelu"),
                                                    (2, 256, 128, 4, 16, "gelu"), (4, 128, 64, 8, 16, "relu")])
def test_conv_dgrad_with_fused_groupnorm_backward(dev, n, c, c0, hw, groups, act):
    """Data-gradient convolution whose loader applies the GroupNorm + activation backward (2x2 / 4x4 / 8x8 images): the raw
    gradient, the per-image parameter-gradient rows and the data-gradient must match the two-launch path (same kernels'
    arithmetic, different reduction order): 1e-2 of the scale for bf16 tensors, 2e-3 relative for the fp32 rows."""
    raw = r16(hash_normal((n, hw, hw, c), 101) * 1.3 + 0.1).to(dev).bfloat16()
    dy = r16(hash_normal((n, hw, hw, c), 102)).to(dev).bfloat16()
    gamma, beta = (1 + 0.3 * hash_normal((c,), 103)).to(dev), (0.3 * hash_normal((c,), 104)).to(dev)
    w = r16(hash_normal((c, c0, 3, 3), 105) * (1.0 / (3 * c ** 0.5)))
    wd = pack.pack_dgrad(hip.CONV_3X3_S1, w).to(dev)
    res = r16(hash_normal((n, hw, hw, c0), 106)).to(dev).bfloat16()
    _, stats = ops.gn_act_forward(raw, gamma, beta, groups, hip.ACT[act])
    # two launches
    part0 = torch.zeros(n, 3, c, device=dev)
    draw0 = ops.gn_act_backward(raw, dy, gamma, beta, groups, hip.ACT[act], stats, part0)
    dsum0 = torch.zeros(n, c0, device=dev)
    dx0 = ops.conv_forward(hip.CONV_3X3_S1, draw0, wd, None, c0, res=res, rows=c0, sum_out=dsum0)
    # one launch
    assert ops.conv_gnb_fusable(hip.CONV_3X3_S1, dy, c0, groups)
    part1 = torch.full((n, 3, c), float("nan"), device=dev)
    draw1 = torch.full_like(raw, float("nan"))
    dsum1 = torch.zeros(n, c0, device=dev)
    dx1 = ops.conv_forward(hip.CONV_3X3_S1, dy, wd, None, c0, res=res, rows=c0, sum_out=dsum1,
                           gnb=(raw, gamma, beta, groups, hip.ACT[act], stats, part1, draw1))
    assert_rel(draw1.float().cpu(), draw0.float().cpu())
    assert_rel(dx1.float().cpu(), dx0.float().cpu())
    torch.testing.assert_close(part1.cpu(), part0.cpu(), rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(dsum1.cpu(), dsum0.cpu(), rtol=1e-2, atol=2e-2)
    assert not ops.conv_gnb_fusable(hip.CONV_3X3_S1, torch.empty(2, 64, 64, 32, device=dev, dtype=torch.bfloat16), 32, 16)


def _fwd_dgrad_check(dev, kind, n, ci, co, h, w_, seed=0):
    k = {hip.CONV_3X3_S1: 3, hip.CONV_3X3_S2: 3, hip.CONV_1X1: 1, hip.CONVT_2X2_S2: 2}[kind]
    wshape = (ci, co, k, k) if kind == hip.CONVT_2X2_S2 else (co, ci, k, k)
    w = r16(hash_normal(wshape, 11 + seed) * (1.0 / (ci * k * k)) ** 0.5)
    b = hash_normal((co,), 12 + seed) * 0.1
    x = r16(hash_normal((n, ci, h, w_), 13 + seed))
    ref = torch_ref(kind, x, w, b)
    y = ops.conv_forward(kind, nhwc(x, dev), pack.pack_fwd(kind, w.to(dev)), b.to(dev), co)
    tile_fwd = hip.lib().ddpm_conv_debug_tile(-1)
    assert_rel(nchw(y), ref)
    gy = r16(hash_normal(tuple(ref.shape), 14 + seed))
    xg = x.clone().requires_grad_(True)
    (gref,) = torch.autograd.grad(torch_ref(kind, xg, w, None), xg, gy)
    gx = ops.conv_forward(pack.DGRAD_KIND[kind], nhwc(gy, dev), pack.pack_dgrad(kind, w.to(dev)), None, ci)
    tile_bwd = hip.lib().ddpm_conv_debug_tile(-1)
    assert_rel(nchw(gx), gref)
    return tile_fwd, tile_bwd


# every workgroup tile of the general kernel (wm x wn waves, nt column blocks per wave), forced on small and ragged problems
TILE_CASES = [
    # tile, kind, N, Cin, Cout, H, W   (Cin and Cout multiples of the tile's channel width: the data-gradient swaps them)
    (414, hip.CONV_1X1, 3, 128, 128, 12, 20),
    (414, hip.CONVT_2X2_S2, 2, 128, 128, 6, 10),
    (414, hip.CONV_3X3_S2, 2, 128, 128, 12, 12),       # data-gradient only (four parity classes of <= 4 taps)
    (412, hip.CONV_3X3_S1, 2, 64, 64, 12, 20),
    (412, hip.CONV_3X3_S2, 3, 64, 64, 16, 16),
    (412, hip.CONV_1X1, 5, 128, 64, 8, 8),
    (412, hip.CONVT_2X2_S2, 2, 64, 64, 8, 8),
    (222, hip.CONV_1X1, 3, 128, 256, 12, 20),
    (222, hip.CONVT_2X2_S2, 3, 256, 128, 4, 4),
    (222, hip.CONV_3X3_S2, 3, 128, 128, 8, 8),         # data-gradient only
    (411, hip.CONV_3X3_S1, 2, 96, 32, 12, 20),
    (221, hip.CONV_3X3_S1, 3, 64, 64, 10, 10),
    (221, hip.CONV_3X3_S2, 2, 64, 128, 8, 8),
]


@pytest.mark.parametrize("tile,kind,n,ci,co,h,w_", TILE_CASES)
def test_conv_every_tile_variant(dev, tile, kind, n, ci, co, h, w_):
    """The general kernel picks its tile from the launch size, so the tiles the benchmarked batches run on are forced here on
    problems the CPU reference finishes quickly: forward and data-gradient against torch fp32 on the same bf16 operands,
    1e-2 of the output scale, and the library reports that the forced tile is the one that ran."""
    lib = hip.lib()
    lib.ddpm_conv_debug_tile(tile)
    try:
        three = kind in (hip.CONV_3X3_S1, hip.CONV_3X3_S2)
        if three and tile in (414, 222):
            # nine-tap problems cannot take 128-channel tiles (LDS); the stride-2 data-gradient can: force only that launch
            lib.ddpm_conv_debug_tile(0)
            k = 3
            w = r16(hash_normal((co, ci, k, k), 21) * (1.0 / (ci * 9)) ** 0.5)
            x = r16(hash_normal((n, ci, h, w_), 22))
            ref = torch_ref(kind, x, w, None)
            gy = r16(hash_normal(tuple(ref.shape), 23))
            xg = x.clone().requires_grad_(True)
            (gref,) = torch.autograd.grad(torch_ref(kind, xg, w, None), xg, gy)
            lib.ddpm_conv_debug_tile(tile)
            gx = ops.conv_forward(pack.DGRAD_KIND[kind], nhwc(gy, dev), pack.pack_dgrad(kind, w.to(dev)), None, ci)
            assert lib.ddpm_conv_debug_tile(-1) == tile
            assert_rel(nchw(gx), gref)
            return
        tf, tb = _fwd_dgrad_check(dev, kind, n, ci, co, h, w_)
        assert tf == tile and tb == tile, (tf, tb, tile)
    finally:
        lib.ddpm_conv_debug_tile(0)


@pytest.mark.parametrize("want,kind,n,ci,co,h", [
    (414, hip.CONV_1X1, 64, 256, 128, 32),         # large-batch shapes: the automatic choice reaches each tile (a pointwise shape
                                                   # the streaming kernel does not take)
    (412, hip.CONV_3X3_S1, 128, 64, 64, 32),
    (414, hip.CONVT_2X2_S2, 128, 128, 128, 16),
])
def test_conv_large_launches_pick_the_wide_tiles(dev, want, kind, n, ci, co, h):
    """Launches as large as the benchmarked 64x64 / 128x128 batches choose the wide tiles by themselves; checked against
    torch like every other shape."""
    tf, _ = _fwd_dgrad_check(dev, kind, n, ci, co, h, h, seed=40)
    assert tf == want, (tf, want)


# ---- whole-image tiles: conv -> GroupNorm -> act in one launch for every convolution kind of the Down / Up blocks
IMG_FWD_CASES = [
    # kind, N, Cin, C1 (concatenated), Cout, H (input), groups, act, residual, time-embedding add
    (hip.CONV_3X3_S2, 3, 32, 0, 32, 32, 16, "gelu", False, True),       # encoder stage 0 entry at 32x32 input
    (hip.CONV_3X3_S1, 2, 32, 0, 32, 16, 16, "gelu", True, False),
    (hip.CONV_3X3_S1, 3, 32, 32, 32, 16, 16, "silu", False, True),      # decoder entry: concatenated skip connection
    (hip.CONV_3X3_S2, 3, 32, 0, 64, 16, 16, "gelu", False, True),
    (hip.CONV_3X3_S2, 5, 64, 0, 128, 8, 16, "gelu", False, True),       # two images per 32-row block, ragged batch
    (hip.CONV_3X3_S2, 9, 128, 0, 256, 4, 16, "relu", False, True),      # eight images per block, ragged batch
    (hip.CONVT_2X2_S2, 9, 256, 0, 128, 2, 16, "gelu", False, False),
    (hip.CONVT_2X2_S2, 3, 128, 0, 64, 4, 16, "silu", False, False),
    (hip.CONVT_2X2_S2, 2, 64, 0, 32, 8, 16, "gelu", False, False),
    (hip.CONVT_2X2_S2, 2, 32, 0, 32, 16, 16, "gelu", False, False),     # 32x32 output: 1024 rows per image
    (hip.CONV_3X3_S1, 2, 32, 0, 32, 32, 16, "gelu", False, False),      # the 32x32 stage of a 64x64 input
    (hip.CONV_3X3_S2, 2, 32, 0, 64, 32, 16, "gelu", False, True),
]


@pytest.mark.parametrize("kind,n,ci,c1,co,h,groups,act,with_res,with_add", IMG_FWD_CASES)
def test_conv_img_fused_groupnorm_forward(dev, kind, n, ci, c1, co, h, groups, act, with_res, with_add):
    """conv -> GroupNorm -> act (+ residual) in one launch on whole-image tiles: raw result, statistics and normalised
    output against torch fp32 on the same bf16 operands (1e-2 / 1.5e-2 of the scale; statistics 2e-2 / 1e-2 relative),
    and against the two-launch path."""
    from oracle import ref_cpu as O
    k = 2 if kind == hip.CONVT_2X2_S2 else 3
    x = r16(hash_normal((n, ci, h, h), 161))
    xb = r16(hash_normal((n, c1, h, h), 162)) if c1 else None
    wshape = (ci + c1, co, k, k) if kind == hip.CONVT_2X2_S2 else (co, ci + c1, k, k)
    w = r16(hash_normal(wshape, 163) * (1.5 / (k * (ci + c1) ** 0.5)))
    b = hash_normal((co,), 164) * 0.1
    gamma, beta = 1 + 0.3 * hash_normal((co,), 165), 0.3 * hash_normal((co,), 166)
    add = hash_normal((n, ci), 168) * 0.5 if with_add else None
    xin = x if add is None else r16(x + add[:, :, None, None])
    raw_ref = r16(torch_ref(kind, xin if xb is None else torch.cat([xin, xb], 1), w, b))
    oh = raw_ref.shape[2]
    res = r16(hash_normal((n, co, oh, oh), 167)) if with_res else None
    y_ref = O.act_fn(act)(F.group_norm(raw_ref, groups, gamma, beta, eps=1e-5))
    if with_res:
        y_ref = y_ref + res
    g = raw_ref.reshape(n, groups, -1)
    mean_ref, rstd_ref = g.mean(2), (g.var(2, unbiased=False) + 1e-5).rsqrt()
    x0d, x1d = nhwc(x, dev), (nhwc(xb, dev) if xb is not None else None)
    wp = pack.pack_fwd(kind, w).to(dev)
    assert ops.conv_gn_fusable(kind, x0d, x1d, co, groups)
    kw = dict(x1=x1d, in_add=None if add is None else add.to(dev))
    raw, y, stats = ops.conv_forward(kind, x0d, wp, b.to(dev), co, gn=(gamma.to(dev), beta.to(dev), groups, hip.ACT[act],
                                                                       None if res is None else nhwc(res, dev)), **kw)
    assert_rel(nchw(raw), raw_ref)
    assert_rel(nchw(y), y_ref, 1.5e-2)
    torch.testing.assert_close(stats[..., 0].cpu(), mean_ref, rtol=2e-2, atol=2e-3)
    torch.testing.assert_close(stats[..., 1].cpu(), rstd_ref, rtol=1e-2, atol=1e-3)
    raw2 = ops.conv_forward(kind, x0d, wp, b.to(dev), co, **kw)
    y2, stats2 = ops.gn_act_forward(raw2, gamma.to(dev), beta.to(dev), groups, hip.ACT[act], res=None if res is None else nhwc(res, dev))
    assert_rel(raw.float().cpu(), raw2.float().cpu(), 4e-3)              # other summation order: at most one bf16 step
    torch.testing.assert_close(stats, stats2, rtol=2e-3, atol=2e-3)
    assert_rel(y.float().cpu(), y2.float().cpu(), 1.5e-2)
    # inference: the raw result is not stored, the normalised one and the statistics are the same bits
    none, y3, stats3 = ops.conv_forward(kind, x0d, wp, b.to(dev), co, keep_raw=False,
                                        gn=(gamma.to(dev), beta.to(dev), groups, hip.ACT[act], None if res is None else nhwc(res, dev)), **kw)
    assert none is None and torch.equal(y3, y) and torch.equal(stats3, stats)


IMG_BWD_CASES = [
    # forward kind of the unit, N, unit input channels (data-gradient width), unit output channels, H of the unit's OUTPUT, groups, act
    (hip.CONV_3X3_S1, 2, 32, 32, 16, 16, "gelu"),
    (hip.CONV_3X3_S1, 2, 32, 32, 32, 16, "gelu"),
    (hip.CONV_3X3_S2, 2, 32, 32, 16, 16, "gelu"),       # dY 16x16 -> dX 32x32: four parity classes, 1024 rows per image
    (hip.CONV_3X3_S2, 3, 32, 64, 8, 16, "silu"),
    (hip.CONV_3X3_S2, 5, 64, 128, 4, 16, "gelu"),
    (hip.CONVT_2X2_S2, 9, 256, 128, 4, 16, "gelu"),     # dY 4x4 -> dX 2x2: eight images per block, ragged batch
    (hip.CONVT_2X2_S2, 3, 128, 64, 8, 16, "relu"),
    (hip.CONVT_2X2_S2, 2, 64, 32, 16, 16, "gelu"),
    (hip.CONVT_2X2_S2, 2, 32, 32, 32, 16, "gelu"),      # dY 32x32: the loader walks 64 chunks per image
]


@pytest.mark.parametrize("kind,n,c0,c,hw,groups,act", IMG_BWD_CASES)
def test_conv_img_dgrad_with_fused_groupnorm_backward(dev, kind, n, c0, c, hw, groups, act):
    """Data-gradient launch whose loader applies the unit's GroupNorm + activation backward, for every convolution kind: raw
    gradient, per-image parameter-gradient rows, data-gradient and its per-image sums against the two-launch path (same
    arithmetic, other summation order): 1e-2 of the scale for bf16 tensors, 2e-3 for the fp32 rows."""
    k = 2 if kind == hip.CONVT_2X2_S2 else 3
    dk = pack.DGRAD_KIND[kind]
    raw = r16(hash_normal((n, hw, hw, c), 201) * 1.3 + 0.1).to(dev).bfloat16()
    dy = r16(hash_normal((n, hw, hw, c), 202)).to(dev).bfloat16()
    gamma, beta = (1 + 0.3 * hash_normal((c,), 203)).to(dev), (0.3 * hash_normal((c,), 204)).to(dev)
    wshape = (c0, c, k, k) if kind == hip.CONVT_2X2_S2 else (c, c0, k, k)
    w = r16(hash_normal(wshape, 205) * (1.0 / (k * c ** 0.5)))
    wd = pack.pack_dgrad(kind, w).to(dev)
    ih = {hip.CONV_3X3_S1: hw, hip.CONV_3X3_S2: 2 * hw, hip.CONVT_2X2_S2: hw // 2}[kind]
    res = r16(hash_normal((n, ih, ih, c0), 206)).to(dev).bfloat16()
    _, stats = ops.gn_act_forward(raw, gamma, beta, groups, hip.ACT[act])
    part0 = torch.zeros(n, 3, c, device=dev)
    draw0 = ops.gn_act_backward(raw, dy, gamma, beta, groups, hip.ACT[act], stats, part0)
    dsum0 = torch.zeros(n, c0, device=dev)
    dx0 = ops.conv_forward(dk, draw0, wd, None, c0, res=res, rows=c0, sum_out=dsum0)
    assert ops.conv_gnb_fusable(dk, dy, c0, groups)
    part1 = torch.full((n, 3, c), float("nan"), device=dev)
    draw1 = torch.full_like(raw, float("nan"))
    dsum1 = torch.zeros(n, c0, device=dev)
    dx1 = ops.conv_forward(dk, dy, wd, None, c0, res=res, rows=c0, sum_out=dsum1,
                           gnb=(raw, gamma, beta, groups, hip.ACT[act], stats, part1, draw1))
    assert_rel(draw1.float().cpu(), draw0.float().cpu())
    assert_rel(dx1.float().cpu(), dx0.float().cpu())
    torch.testing.assert_close(part1.cpu(), part0.cpu(), rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(dsum1.cpu(), dsum0.cpu(), rtol=1e-2, atol=2e-2 * max(1.0, hw / 8))


@pytest.mark.parametrize("kind,n,ci,co,h,wd,groups,act", [
    (hip.CONV_3X3_S1, 2, 32, 32, 64, 64, 4, "gelu"),       # 8 channels per group, several tiles per image
    (hip.CONV_1X1, 3, 32, 128, 64, 64, 4, "silu"),         # expansion of an inverted-residual block
    (hip.CONV_1X1, 1, 128, 32, 128, 128, 4, "none"),       # projection
    (hip.CONVT_2X2_S2, 2, 64, 32, 32, 32, 4, "gelu"),      # four position classes
    (hip.CONV_3X3_S1, 1, 64, 64, 64, 128, 8, "relu"),      # non-square
    (hip.CONV_1X1, 2, 32, 64, 128, 128, 16, "gelu"),       # 4 channels per group (the UNet's 16 groups at 64 channels)
    (hip.CONV_1X1, 2, 64, 32, 64, 64, 16, "none"),         # 2 channels per group
    (hip.CONV_3X3_S1, 1, 32, 256, 64, 64, 4, "gelu"),      # 64 channels per group
    (hip.CONV_1X1, 5, 128, 256, 32, 32, 16, "gelu"),       # 32x32 images wide enough for the many-workgroup GroupNorm
    (hip.CONV_3X3_S1, 3, 64, 64, 32, 32, 16, "silu"),
])
def test_conv_leaves_groupnorm_sums_for_large_images(dev, kind, n, ci, co, h, wd, groups, act):
    """Large images: every output tile of the convolution stores the sums of its results per normalisation group and the
    GroupNorm that follows runs without its statistics pass.  The raw tensor is bit-equal to the plain launch; statistics
    and output equal those of the two-pass GroupNorm up to fp32 summation order (1e-5 relative / one bf16 step), and
    torch's F.group_norm within the bf16 tolerance."""
    from oracle import ref_cpu as O
    k = {hip.CONV_3X3_S1: 3, hip.CONV_1X1: 1, hip.CONVT_2X2_S2: 2}[kind]
    wshape = (ci, co, k, k) if kind == hip.CONVT_2X2_S2 else (co, ci, k, k)
    w = r16(hash_normal(wshape, 81) * (1.0 / (ci * k * k)) ** 0.5)
    b = hash_normal((co,), 82) * 0.3
    x = r16(hash_normal((n, ci, h, wd), 83))
    gamma, beta = (1 + 0.3 * hash_normal((co,), 84)).to(dev), (0.3 * hash_normal((co,), 85)).to(dev)
    xd, wp, bd = nhwc(x, dev), pack.pack_fwd(kind, w.to(dev)), b.to(dev)
    raw, part = ops.conv_forward(kind, xd, wp, bd, co, gn_part_groups=groups)
    assert part is not None and part.shape[:2] == (n, groups)
    raw2 = ops.conv_forward(kind, xd, wp, bd, co)
    assert torch.equal(raw, raw2)
    # the partial sums add up to the sums of the stored tensor
    g = raw.float().view(n, -1, groups, co // groups)
    torch.testing.assert_close(part[..., 0].sum(2), g.sum((1, 3)), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(part[..., 1].sum(2), (g * g).sum((1, 3)), rtol=1e-4, atol=1e-2)
    y, stats = ops.gn_act_forward(raw, gamma, beta, groups, hip.ACT[act], stat_part=part)
    y2, stats2 = ops.gn_act_forward(raw2, gamma, beta, groups, hip.ACT[act])
    torch.testing.assert_close(stats, stats2, rtol=1e-5, atol=1e-6)
    assert_rel(y.float().cpu(), y2.float().cpu(), 8e-3)
    y_ref = F.group_norm(nchw(raw), groups, gamma.cpu(), beta.cpu(), eps=1e-5)
    assert_rel(nchw(y), y_ref if act == "none" else O.act_fn(act)(y_ref), 1e-2)


def test_conv_groupnorm_sums_are_declined_where_they_do_not_apply(dev):
    """small images (whole-image kernels / single-pass GroupNorm) and one-channel groups get no partial sums"""
    w = pack.pack_fwd(hip.CONV_1X1, torch.zeros(32, 32, 1, 1, device=dev))
    for shape, groups in (((2, 16, 16, 32), 4), ((1, 64, 64, 32), 32), ((1, 60, 68, 32), 4)):
        x = torch.zeros(shape, device=dev, dtype=torch.bfloat16)
        raw, part = ops.conv_forward(hip.CONV_1X1, x, w, None, 32, gn_part_groups=groups)
        assert part is None and raw.shape == shape


@pytest.mark.parametrize("n,ci,co,h,wd,groups,act,with_sums", [
    (2, 64, 32, 128, 128, 16, "gelu", True),        # projection of an inverted-residual block at the first stage
    (3, 128, 64, 64, 64, 16, "silu", True),
    (2, 64, 32, 16, 16, 16, "gelu", False),         # small image: no partial sums for the output, the loader still normalises
    (1, 256, 128, 64, 64, 16, "none", True),
])
def test_conv_normalises_and_activates_on_load(dev, n, ci, co, h, wd, groups, act, with_sums):
    """Inference on large images: a 1x1 convolution reads the RAW result of the producer and applies its GroupNorm + activation
    in the loader (scale / shift per image and channel from the producer's partial sums).  Against the three-launch path
    (GroupNorm launch, then the convolution): the statistics agree to 1e-5, the result within one bf16 step of its scale (the
    two paths add the partial sums in different orders), and against torch fp32 within the bf16 tolerance."""
    from oracle import ref_cpu as O
    x = r16(hash_normal((n, ci, h, wd), 91) * 1.3 + 0.4)
    gamma, beta = (1 + 0.3 * hash_normal((ci,), 92)), (0.3 * hash_normal((ci,), 93))
    w = r16(hash_normal((co, ci, 1, 1), 94) * (1.0 / ci) ** 0.5)
    b = hash_normal((co,), 95) * 0.2
    xd, wp, bd = nhwc(x, dev), pack.pack_fwd(hip.CONV_1X1, w.to(dev)), b.to(dev)
    # the producer's partial sums: here from a plain reduction of x over arbitrary chunks of pixels
    parts = 8
    g = xd.float().view(n, parts, h * wd // parts, groups, ci // groups)
    part = torch.stack([g.sum((2, 4)), (g * g).sum((2, 4))], -1).permute(0, 2, 1, 3).contiguous()     # [N][G][parts][2]
    stats, affine = ops.gn_fold_affine(part, gamma.to(dev), beta.to(dev), h * wd)
    y_gn, stats2 = ops.gn_act_forward(xd, gamma.to(dev), beta.to(dev), groups, hip.ACT[act])
    torch.testing.assert_close(stats, stats2, rtol=1e-5, atol=1e-6)
    ref = ops.conv_forward(hip.CONV_1X1, y_gn, wp, bd, co)
    got = ops.conv_forward(hip.CONV_1X1, xd, wp, bd, co, in_affine=(affine, hip.ACT[act]), gn_part_groups=groups if with_sums else 0)
    if with_sums:
        got, opart = got
        assert opart is not None
        gg = got.float().view(n, -1, groups, co // groups)
        torch.testing.assert_close(opart[..., 0].sum(2), gg.sum((1, 3)), rtol=1e-4, atol=1e-2)
    assert_rel(got.float().cpu(), ref.float().cpu(), 8e-3)
    a = (lambda v: v) if act == "none" else O.act_fn(act)
    y_ref = F.conv2d(r16(a(F.group_norm(x, groups, gamma, beta, eps=1e-5))), w, b)
    assert_rel(nchw(got), y_ref, 1e-2)


@pytest.mark.parametrize("n,ci,co,h,wd,groups,act", [
    (2, 32, 64, 128, 128, 16, None),           # expansion at the first stage: plain input, sums for the GroupNorm that follows
    (3, 64, 32, 64, 64, 16, "gelu"),           # projection: the depthwise result normalised and activated on load
    (2, 64, 128, 64, 32, 16, None),
    (1, 128, 64, 32, 48, 4, "silu"),           # 16 channels per group; image area 1536 = 12 x 128 positions
    (5, 32, 32, 32, 32, 16, "none"),
    (2, 128, 32, 64, 64, 16, "gelu"),
    (2, 32, 128, 16, 24, 8, None),             # three workgroups of 128 positions per image
    (2, 64, 64, 128, 64, 32, "gelu"),          # 2 channels per group
])
def test_pointwise_streaming_kernel(dev, n, ci, co, h, wd, groups, act):
    """The streaming kernel of pointwise convolutions (weights in registers, wave-private output tiles), forced onto small
    problems: the raw result against the general kernel on the same operands (both round one fp32 dot product per element, in a
    different order: one bf16 step of the tensor's scale) and against torch fp32; the per-workgroup GroupNorm sums against the
    sums of the stored tensor, and the GroupNorm that consumes them against the two-pass GroupNorm."""
    from oracle import ref_cpu as O
    lib = hip.lib()
    x = r16(hash_normal((n, ci, h, wd), 191) * 1.2 + 0.3)
    w = r16(hash_normal((co, ci, 1, 1), 192) * (1.0 / ci) ** 0.5)
    b = hash_normal((co,), 193) * 0.3
    gamma, beta = (1 + 0.3 * hash_normal((ci,), 194)), (0.3 * hash_normal((ci,), 195))
    xd, wp, bd = nhwc(x, dev), pack.pack_fwd(hip.CONV_1X1, w.to(dev)), b.to(dev)
    in_aff = None
    x_eff = x
    if act is not None:
        gin = 16
        g = xd.float().view(n, 4, h * wd // 4, gin, ci // gin)
        part_in = torch.stack([g.sum((2, 4)), (g * g).sum((2, 4))], -1).permute(0, 2, 1, 3).contiguous()
        _, affine = ops.gn_fold_affine(part_in, gamma.to(dev), beta.to(dev), h * wd)
        in_aff = (affine, hip.ACT[act] if act != "none" else 0)
        a = (lambda v: v) if act == "none" else O.act_fn(act)
        x_eff = r16(a(F.group_norm(x, gin, gamma, beta, eps=1e-5)))
    prev = lib.ddpm_conv_pw_stream_mode(1)
    try:
        got, part = ops.conv_forward(hip.CONV_1X1, xd, wp, bd, co, in_affine=in_aff, gn_part_groups=groups)
        assert lib.ddpm_conv_last_kernel() == 6
        plain = ops.conv_forward(hip.CONV_1X1, xd, wp, bd, co, in_affine=in_aff)
        assert lib.ddpm_conv_last_kernel() == 6
        lib.ddpm_conv_pw_stream_mode(-1)
        ref = ops.conv_forward(hip.CONV_1X1, xd, wp, bd, co, in_affine=in_aff)
        assert lib.ddpm_conv_last_kernel() == 1
    finally:
        lib.ddpm_conv_pw_stream_mode(prev)
    assert torch.equal(got, plain)
    assert_rel(got.float().cpu(), ref.float().cpu(), 8e-3)
    assert_rel(nchw(got), F.conv2d(x_eff, w, b), 1e-2)
    if h * wd * (co // 8 if co < 64 else 8) > 4096 and h > 8 and wd > 8:
        assert part is not None and part.shape[:2] == (n, groups) and (h * wd) % part.shape[2] == 0
        gg = got.float().view(n, -1, groups, co // groups)
        torch.testing.assert_close(part[..., 0].sum(2), gg.sum((1, 3)), rtol=1e-4, atol=1e-2)
        torch.testing.assert_close(part[..., 1].sum(2), (gg * gg).sum((1, 3)), rtol=1e-4, atol=1e-2)
        go, bo = (1 + 0.3 * hash_normal((co,), 196)).to(dev), (0.3 * hash_normal((co,), 197)).to(dev)
        y, stats = ops.gn_act_forward(got, go, bo, groups, hip.ACT["gelu"], stat_part=part)
        y2, stats2 = ops.gn_act_forward(got, go, bo, groups, hip.ACT["gelu"])
        torch.testing.assert_close(stats, stats2, rtol=1e-5, atol=1e-6)
        assert_rel(y.float().cpu(), y2.float().cpu(), 8e-3)
    else:
        assert part is None


@pytest.mark.parametrize("n,ci,co,h,act", [(128, 32, 64, 128, None), (64, 64, 32, 128, "gelu"), (160, 64, 128, 64, None), (256, 128, 64, 32, "silu")])
def test_pointwise_streaming_kernel_at_sampling_batch(dev, n, ci, co, h, act):
    """Sizes the dispatcher sends there by itself (32 K positions and more; these have workgroups of 128 .. 1024 positions), against
    the general kernel on the device: result within one bf16 step of the scale, partial sums against the stored tensor."""
    lib = hip.lib()
    gen = torch.Generator(device=dev).manual_seed(7)
    xd = (torch.randn(n, h, h, ci, device=dev, generator=gen) * 1.1 + 0.2).bfloat16()
    wp = pack.pack_fwd(hip.CONV_1X1, torch.randn(co, ci, 1, 1, device=dev, generator=gen) / ci ** 0.5)
    bd = torch.randn(co, device=dev, generator=gen) * 0.3
    in_aff = None
    if act is not None:
        affine = torch.stack([torch.rand(n, ci, device=dev, generator=gen) + 0.5, torch.randn(n, ci, device=dev, generator=gen) * 0.3], 2).contiguous()
        in_aff = (affine, hip.ACT[act])
    got, part = ops.conv_forward(hip.CONV_1X1, xd, wp, bd, co, in_affine=in_aff, gn_part_groups=16)
    assert lib.ddpm_conv_last_kernel() == 6 and part is not None and part.shape[2] <= h * h // 128
    prev = lib.ddpm_conv_pw_stream_mode(-1)
    try:
        ref = ops.conv_forward(hip.CONV_1X1, xd, wp, bd, co, in_affine=in_aff)
    finally:
        lib.ddpm_conv_pw_stream_mode(prev)
    scale = ref.float().abs().max().item()
    assert (got.float() - ref.float()).abs().max().item() <= 8e-3 * scale
    gg = got.float().view(n, -1, 16, co // 16)
    torch.testing.assert_close(part[..., 0].sum(2), gg.sum((1, 3)), rtol=1e-4, atol=5e-2)
    torch.testing.assert_close(part[..., 1].sum(2), (gg * gg).sum((1, 3)), rtol=1e-4, atol=5e-2)


@pytest.mark.parametrize("n,ci,co,h,act", [(2, 32, 64, 64, None), (3, 64, 32, 32, "gelu"), (1, 128, 64, 16, None)])
def test_pointwise_streaming_kernel_with_input_dropout(dev, n, ci, co, h, act):
    """the block input's dropout mask (one keep bit per element, applied after the normalise-on-load) in the streaming kernel:
    bit-equal to the general kernel's mask handling up to the dot product's rounding, and against torch fp32"""
    from oracle import ref_cpu as O
    lib = hip.lib()
    x = r16(hash_normal((n, ci, h, h), 291))
    w = r16(hash_normal((co, ci, 1, 1), 292) * (1.0 / ci) ** 0.5)
    b = hash_normal((co,), 293) * 0.3
    keep = hash_uniform((n, h, h, ci), 294) < 0.8
    kb = (keep.reshape(-1, 8).to(torch.uint8) << torch.arange(8, dtype=torch.uint8)).sum(1).to(torch.uint8).to(dev)
    xd, wp, bd = nhwc(x, dev), pack.pack_fwd(hip.CONV_1X1, w.to(dev)), b.to(dev)
    in_aff, x_eff = None, x
    if act is not None:
        sc, sh = hash_uniform((n, ci), 295) + 0.5, hash_normal((n, ci), 296) * 0.3
        in_aff = (torch.stack([sc, sh], 2).contiguous().to(dev), hip.ACT[act])
        x_eff = r16(O.act_fn(act)(x * sc[:, :, None, None] + sh[:, :, None, None]))
    x_eff = r16(x_eff * keep.permute(0, 3, 1, 2) / 0.8)
    prev = lib.ddpm_conv_pw_stream_mode(1)
    try:
        got = ops.conv_forward(hip.CONV_1X1, xd, wp, bd, co, in_affine=in_aff, in_mask=kb, in_scale=1 / 0.8)
        assert lib.ddpm_conv_last_kernel() == 6
        lib.ddpm_conv_pw_stream_mode(-1)
        ref = ops.conv_forward(hip.CONV_1X1, xd, wp, bd, co, in_affine=in_aff, in_mask=kb, in_scale=1 / 0.8)
    finally:
        lib.ddpm_conv_pw_stream_mode(prev)
    assert_rel(got.float().cpu(), ref.float().cpu(), 8e-3)
    assert_rel(nchw(got), F.conv2d(x_eff, w, b), 1e-2)


@pytest.mark.parametrize("n,ci,co,h,with_mask,with_res", [(2, 64, 32, 64, True, True), (3, 32, 64, 32, True, False), (1, 128, 64, 16, False, True),
                                                           (2, 64, 64, 32, True, True)])
def test_pointwise_streaming_kernel_masked_and_added_result(dev, n, ci, co, h, with_mask, with_res):
    """the data-gradient form of the block's expansion (conv_block.py:57-76 of the reference, backward): the result's dropout mask and
    the tensor added last.  The general kernel applies both to the bf16-rounded product, and so does this one: the two differ by
    the dot product's rounding only; and against torch fp32."""
    lib = hip.lib()
    x = r16(hash_normal((n, ci, h, h), 391))
    w = r16(hash_normal((co, ci, 1, 1), 392) * (1.0 / ci) ** 0.5)
    res = r16(hash_normal((n, co, h, h), 393)) if with_res else None
    keep = hash_uniform((n, h, h, co), 394) < 0.8
    kb = (keep.reshape(-1, 8).to(torch.uint8) << torch.arange(8, dtype=torch.uint8)).sum(1).to(torch.uint8).to(dev) if with_mask else None
    xd, wp = nhwc(x, dev), pack.pack_fwd(hip.CONV_1X1, w.to(dev))
    kw = dict(res=None if res is None else nhwc(res, dev), out_mask=kb, out_scale=1 / 0.8 if with_mask else 1.0)
    prev = lib.ddpm_conv_pw_stream_mode(1)
    try:
        got = ops.conv_forward(hip.CONV_1X1, xd, wp, None, co, **kw)
        assert lib.ddpm_conv_last_kernel() == 6
        lib.ddpm_conv_pw_stream_mode(-1)
        ref = ops.conv_forward(hip.CONV_1X1, xd, wp, None, co, **kw)
        assert lib.ddpm_conv_last_kernel() != 6
    finally:
        lib.ddpm_conv_pw_stream_mode(prev)
    assert_rel(got.float().cpu(), ref.float().cpu(), 8e-3)
    want = F.conv2d(x, w)
    if with_mask:
        want = want * keep.permute(0, 3, 1, 2) / 0.8
        if not with_res:
            assert torch.equal((got == 0).cpu() | keep, torch.ones_like(keep))      # dropped elements are exact zeros
    if with_res:
        want = want + res
    assert_rel(nchw(got), want, 1e-2)


def test_pointwise_streaming_kernel_declines_what_it_does_not_cover(dev):
    """other shapes, fused epilogues and small problems stay on the other kernels (mode 0 = by size)"""
    lib = hip.lib()
    prev = lib.ddpm_conv_pw_stream_mode(1)
    try:
        for ci, co, h, kw in ((256, 64, 16, {}), (128, 128, 16, {}), (64, 32, 10, {}), (64, 32, 16, {"act": hip.ACT["gelu"]}),
                              (64, 32, 16, {"sum_out": True})):
            x = torch.zeros(2, h, h, ci, device=dev, dtype=torch.bfloat16)
            w = pack.pack_fwd(hip.CONV_1X1, torch.zeros(co, ci, 1, 1, device=dev))
            if kw.get("sum_out"):
                kw = {"sum_out": torch.zeros(2, co, device=dev)}
            ops.conv_forward(hip.CONV_1X1, x, w, None, co, **kw)
            assert lib.ddpm_conv_last_kernel() != 6, (ci, co, h, kw)
        lib.ddpm_conv_pw_stream_mode(0)
        x = torch.zeros(2, 64, 64, 32, device=dev, dtype=torch.bfloat16)
        ops.conv_forward(hip.CONV_1X1, x, pack.pack_fwd(hip.CONV_1X1, torch.zeros(64, 32, 1, 1, device=dev)), None, 64)
        assert lib.ddpm_conv_last_kernel() != 6            # 8 K positions: below the size rule (32 K)
    finally:
        lib.ddpm_conv_pw_stream_mode(prev)


PIPE_CASES = [
    # kind, N, C0, C1, Cout, W, act, with residual
    (hip.CONV_3X3_S1, 3, 128, 0, 128, 16, "none", False),
    (hip.CONV_3X3_S1, 5, 256, 0, 256, 8, "gelu", True),        # 4 images per tile, ragged
    (hip.CONV_3X3_S1, 19, 256, 0, 512, 4, "none", False),      # 16 images per tile, ragged; four column tiles
    (hip.CONV_3X3_S1, 17, 512, 0, 256, 4, "silu", False),
    (hip.CONV_3X3_S1, 2, 128, 128, 128, 16, "none", True),     # two concatenated sources
    (hip.CONV_3X3_S1, 6, 64, 0, 128, 8, "none", False),        # a single 64-channel chunk: 9 steps
    (hip.CONV_1X1, 3, 128, 0, 256, 8, "gelu", False),
    (hip.CONV_1X1, 9, 512, 0, 256, 4, "none", True),
    (hip.CONV_1X1, 2, 64, 0, 128, 16, "none", False),
    (hip.CONV_3X3_S1, 7, 128, 128, 128, 8, "add", False),      # decoder form: (Z + time embedding) ++ skip
    (hip.CONV_1X1, 3, 128, 0, 128, 8, "add", False),
]


@pytest.mark.parametrize("kind,n,c0,c1,co,w_,act,with_res", PIPE_CASES)
def test_pipelined_wide_kernel_forward_and_dgrad(dev, kind, n, c0, c1, co, w_, act, with_res):
    """The matrix-core-bound kernel (256 x 128 tiles, three-deep weight pipeline), forced onto small problems: forward with
    bias / activation / residual / two sources, and the data-gradient (the same convolution on the flipped, transposed
    weight copy), against torch fp32 on the same bf16 operands."""
    from oracle import ref_cpu as O
    ci = c0 + c1
    k = 3 if kind == hip.CONV_3X3_S1 else 1
    w = r16(hash_normal((co, ci, k, k), 1) * (1.0 / (ci * k * k)) ** 0.5)
    b = hash_normal((co,), 2) * 0.1
    x = r16(hash_normal((n, ci, w_, w_), 3))
    res = r16(hash_normal((n, co, w_, w_), 5))
    temb = None
    if act == "add":                           # broadcast add on the first source, rounded to bf16 as the loader rounds it
        temb = hash_normal((n, c0), 6)
        x = torch.cat([r16(x[:, :c0] + temb.view(n, c0, 1, 1)), x[:, c0:]], 1)
        act = "none"
    ref = torch_ref(kind, x, w, b)
    if act != "none":
        ref = O.act_fn(act)(ref)
    if with_res:
        ref = ref + res
    prev = hip.lib().ddpm_conv_pipe_mode(1)
    try:
        xd = nhwc(x, dev)
        x0, x1 = (xd[..., :c0].contiguous(), xd[..., c0:].contiguous()) if c1 else (xd, None)
        if temb is not None:                   # hand the kernel the tensor WITHOUT the add
            x0 = (x0.float() - temb.to(dev).view(n, 1, 1, c0)).bfloat16()
            x_chk = (x0.float() + temb.to(dev).view(n, 1, 1, c0)).bfloat16()
            ref = torch_ref(kind, torch.cat([nchw(x_chk), nchw(x1)], 1) if c1 else nchw(x_chk), w, b)
        y = ops.conv_forward(kind, x0, pack.pack_fwd(kind, w.to(dev)), b.to(dev), co, x1=x1, act=hip.ACT.get(act, 0) if act != "none" else 0,
                             res=nhwc(res, dev) if with_res else None, in_add=None if temb is None else temb.to(dev))
        assert hip.lib().ddpm_conv_last_kernel() == 5, "the pipelined kernel did not take the launch"
        assert_rel(nchw(y), ref)
        if ci % 128 == 0 and co % 64 == 0:
            gy = r16(hash_normal((n, co, w_, w_), 4))
            xg = x.clone().requires_grad_(True)
            (gref,) = torch.autograd.grad(torch_ref(kind, xg, w, None), xg, gy)
            gx = ops.conv_forward(pack.DGRAD_KIND[kind], nhwc(gy, dev), pack.pack_dgrad(kind, w.to(dev)), None, ci)
            assert hip.lib().ddpm_conv_last_kernel() == 5
            assert_rel(nchw(gx), gref)
    finally:
        hip.lib().ddpm_conv_pipe_mode(prev)
