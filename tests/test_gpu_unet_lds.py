"""GPU parity of the LDS-resident UNet kernel (ddpm_unet_lds_run), operation by operation, through the C ABI: each
operation of its program format against plain torch fp32 on the same bf16 inputs (1e-2 of the output's scale: bf16
storage, fp32 accumulation), at the tile shapes the denoiser's stages use, with ragged batches and several images per
workgroup.  The whole-network comparison lives in test_gpu_unet.py / test_gpu_configs.py (the engine takes this path by
default where the shapes fit)."""
import pytest
import torch
import torch.nn.functional as F

from dm_amd.runtime import hip, pack
from dm_amd.runtime.lds_program import K11, K33, K33S2, KT22, LdsProgram
from oracle import ref_cpu as O
from oracle.noise import hash_normal, hash_uniform

pytestmark = pytest.mark.gpu

GELU = hip.ACT["gelu"]


def nhwc(x):      # NCHW fp32 (cpu) -> NHWC bf16 contiguous
    return x.permute(0, 2, 3, 1).contiguous().bfloat16()


def nchw(x):
    return x.float().cpu().permute(0, 3, 1, 2)


def close(got, ref, tol=1e-2):
    err = (got - ref).abs().max().item()
    assert err <= tol * (ref.abs().max().item() + 1e-6), (err, ref.abs().max().item())


@pytest.mark.parametrize("n,ipw,w,c", [(3, 1, 16, 32), (3, 2, 4, 128), (9, 8, 2, 256)])
def test_load_store_addt_roundtrip(dev, n, ipw, w, c):
    x = nhwc(hash_normal((n, c, w, w), 1)).to(dev)
    skip = nhwc(hash_normal((n, c, w, w), 2)).to(dev)
    temb = hash_normal((n, c), 3).to(dev)
    out = torch.full((n, w, w, 2 * c), float("nan"), device=dev, dtype=torch.bfloat16)
    p = LdsProgram(ipw, GELU)
    t = p.tile(w, 2 * c)
    p.load(t, x, col0=0)
    p.load(t, skip, col0=c)
    p.addt(t, temb, c, col0=0)
    p.store(t, out, col0=0, ch=c, gcol0=c)          # swapped halves on the way out
    p.store(t, out, col0=c, ch=c, gcol0=0)
    p.check()
    p.run_blocking(n)
    want0 = (x.float() + temb.view(n, 1, 1, c)).bfloat16()
    assert torch.equal(out[..., c:], want0)
    assert torch.equal(out[..., :c], skip)


CONV_CASES = [
    # kind, cin, cout, w_out, ipw, n
    (K33, 32, 32, 16, 1, 3), (K33, 256, 256, 2, 1, 5), (K33, 128, 128, 4, 1, 3), (K33, 64, 64, 8, 1, 2), (K33, 512, 256, 2, 8, 9),
    (K33, 64, 32, 16, 2, 3), (K33, 256, 512, 2, 4, 5),
    (K33S2, 32, 32, 16, 1, 2), (K33S2, 128, 256, 2, 1, 3), (K33S2, 64, 128, 4, 4, 5),
    (K11, 32, 64, 16, 1, 2), (K11, 512, 256, 2, 1, 3), (K11, 128, 256, 4, 2, 3), (K11, 64, 32, 16, 1, 2),
    (KT22, 256, 128, 4, 1, 3), (KT22, 32, 32, 32, 1, 2), (KT22, 128, 64, 8, 8, 9), (KT22, 64, 32, 16, 2, 3),
]


@pytest.mark.parametrize("kind,cin,cout,w_out,ipw,n", CONV_CASES)
def test_conv_kinds_against_torch(dev, kind, cin, cout, w_out, ipw, n):
    w_in = 2 * w_out if kind == K33S2 else w_out // 2 if kind == KT22 else w_out
    x = hash_normal((n, cin, w_in, w_in), 11).bfloat16().float()
    k = {K33: 3, K33S2: 3, K11: 1, KT22: 2}[kind]
    wshape = (cin, cout, 2, 2) if kind == KT22 else (cout, cin, k, k)
    wt = (hash_normal(wshape, 12) / (cin * k * k) ** 0.5).bfloat16().float()
    bias = hash_normal((cout,), 13) * 0.1
    if kind == KT22:
        ref = F.conv_transpose2d(x, wt, bias, stride=2)
    else:
        ref = F.conv2d(x, wt, bias, stride=2 if kind == K33S2 else 1, padding=k // 2)
    xd = nhwc(x).to(dev)
    wp = pack.pack_fwd(kind, wt).to(dev)
    bd = bias.to(dev)
    out = torch.full((n, w_out, w_out, cout + 8), float("nan"), device=dev, dtype=torch.bfloat16)
    p = LdsProgram(ipw, GELU)
    src = p.tile(w_in, cin)
    dst = p.tile(w_out, cout + 8)
    p.load(src, xd)
    p.conv(kind, src, dst, wp, bd, cin, cout, col0=8)
    p.store(dst, out, col0=8, ch=cout, gcol0=8)
    p.check()
    p.run_blocking(n)
    close(nchw(out[..., 8:]), ref)


@pytest.mark.parametrize("c,groups,w,ipw,n,act,extras", [(64, 16, 16, 1, 3, "gelu", True), (512, 32, 2, 8, 9, "gelu", False),
                                                         (32, 16, 16, 2, 3, "silu", True), (256, 16, 2, 1, 2, "relu", False),
                                                         (128, 16, 4, 4, 5, "gelu", True)])
def test_groupnorm_act_mask_residual_and_records(dev, c, groups, w, ipw, n, act, extras):
    x = (hash_normal((n, c, w, w), 21) * 1.5 + 0.3).bfloat16().float()
    res = hash_normal((n, c, w, w), 22).bfloat16().float()
    gamma, beta = 1 + 0.3 * hash_normal((c,), 23), 0.3 * hash_normal((c,), 24)
    keep = (hash_uniform((n, w, w, c), 25) < 0.9)
    scale = 1.0 / 0.9
    y = O.act_fn(act)(F.group_norm(x, groups, gamma, beta, eps=1e-5))
    if extras:
        y = y * keep.permute(0, 3, 1, 2).float() * scale + res
    bits = torch.zeros(n * w * w * c // 8, dtype=torch.uint8)
    kf = keep.reshape(-1, 8).to(torch.uint8)
    for j in range(8):
        bits |= kf[:, j] << j
    xd, rd = nhwc(x).to(dev), nhwc(res).to(dev)
    out = torch.empty(n, w, w, c, device=dev, dtype=torch.bfloat16)
    raw = torch.full_like(out, float("nan"))
    ysave = torch.full_like(out, float("nan"))
    stats = torch.full((n, groups, 2), float("nan"), device=dev)
    p = LdsProgram(ipw, hip.ACT[act])
    t, r = p.tile(w, c), p.tile(w, c)
    p.load(t, xd)
    p.load(r, rd)
    p.gn(t, gamma.to(dev), beta.to(dev), groups, hip.ACT[act], c, res=r if extras else None, mask=bits.to(dev) if extras else None, mscale=scale,
         save_raw=raw, save_y=ysave, save_stats=stats)
    p.store(t, out)
    p.check()
    p.run_blocking(n)
    close(nchw(out), y, 1.5e-2)
    assert torch.equal(raw, xd) and torch.equal(ysave, out)
    xg = x.view(n, groups, -1)
    mean, var = xg.mean(-1), xg.var(-1, unbiased=False)
    torch.testing.assert_close(stats[..., 0].cpu(), mean, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(stats[..., 1].cpu(), (var + 1e-5).rsqrt(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("c,w,ipw,n", [(64, 16, 1, 3), (512, 2, 4, 5), (128, 8, 1, 2), (256, 4, 8, 9), (64, 16, 2, 3)])
def test_depthwise_and_block_input_mask(dev, c, w, ipw, n):
    x = hash_normal((n, c, w, w), 31).bfloat16().float()
    wt = hash_normal((c, 1, 3, 3), 32) / 3
    bias = hash_normal((c,), 33) * 0.1
    keep = (hash_uniform((n, w, w, c), 34) < 0.8)
    bits = torch.zeros(n * w * w * c // 8, dtype=torch.uint8)
    kf = keep.reshape(-1, 8).to(torch.uint8)
    for j in range(8):
        bits |= kf[:, j] << j
    xm = (x * keep.permute(0, 3, 1, 2).float() * 1.25).bfloat16().float()
    ref = F.conv2d(xm, wt, bias, padding=1, groups=c)
    out = torch.empty(n, w, w, c, device=dev, dtype=torch.bfloat16)
    p = LdsProgram(ipw, GELU)
    a, b = p.tile(w, c), p.tile(w, c)
    p.load(a, nhwc(x).to(dev))
    p.mask(a, b, bits.to(dev), c, 1.25)
    p.dw(b, wt.view(c, 9).t().contiguous().to(dev), bias.to(dev), c)
    p.store(b, out)
    p.check()
    p.run_blocking(n)
    close(nchw(out), ref)


def test_program_checks_reject_bad_tables(dev):
    p = LdsProgram(1, GELU)
    from dm_amd.runtime.lds_program import LdsOutOfSpace
    for _ in range(4):
        p.tile(32, 32)                           # 4 x 80 KB live at once do not fit 160 KB
    with pytest.raises(LdsOutOfSpace):
        p.layout()
    p = LdsProgram(1, GELU)
    t = p.tile(16, 32)
    x = torch.zeros(2, 16, 16, 32, device=dev, dtype=torch.bfloat16)
    p.load(t, x)
    w = torch.zeros(9, 32, 32, device=dev, dtype=torch.bfloat16)
    p.conv(K33, t, t, w, None, 32, 32)           # source = destination
    with pytest.raises(RuntimeError, match="overlap"):
        p.check()
    p = LdsProgram(3, GELU)
    t = p.tile(16, 32)
    p.load(t, x)
    with pytest.raises(RuntimeError, match="images per workgroup"):
        p.check()


@pytest.mark.parametrize("size,c,n,act,ipw,deep", [(32, 1, 5, "gelu", 1, 8), (64, 3, 3, "gelu", 1, 8), (32, 3, 4, "silu", 1, 1), (32, 1, 7, "gelu", 2, 4),
                                                   (16, 3, 3, "gelu", 1, 8), (32, 1, 19, "gelu", 1, 8)])
def test_whole_network_inference_through_lds_matches_layer_by_layer_path_and_oracle(dev, size, c, n, act, ipw, deep):
    """BaseUNet.forward in inference: the engine's one-launch LDS path for the low-resolution stages against its layer-by-layer
    kernels (same weights, same input: equal to a few bf16 roundings) and against the oracle (5e-2, the whole-network drift
    bound of test_gpu_unet.py).  Reference: unets.py:100-132,206-240 there."""
    from test_gpu_unet import build, l2_err
    m, cfg, p = build(dev, c, act, head_gain=0.3)
    m.eval()
    x = hash_normal((n, c, size, size), 51)
    t = (hash_uniform((n,), 52) * 120).long().clamp_(0, 119) + 1
    eng = m.model._get_engine()
    m.model.use_graphs = False
    eng.lds_ipw, eng.lds_ipw_deep = ipw, deep
    with torch.no_grad():
        eng.use_lds = False
        ref_gpu = m(x.to(dev), t.to(dev)).cpu()
        eng.use_lds = True
        eng._lds_s0.clear()
        got = m(x.to(dev), t.to(dev)).cpu()
        plan = eng._lds_first_stage(None, None, size, size, 32)
        assert plan is not None, "the LDS path did not engage"
        print("LDS plan (first stage, first deep stage, images per workgroup there):", plan)
        ref = O.unet_forward(p, cfg, x, t)
    assert l2_err(got, ref_gpu) < 2.5e-2, l2_err(got, ref_gpu)       # (a few bf16 roundings apart; the untrained network amplifies them)
    assert l2_err(got, ref) < 5e-2, l2_err(got, ref)
