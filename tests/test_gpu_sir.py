"""GPU parity of the fused SeparableInvertResidual kernel (one launch per block) against a torch fp32 restatement of
conv_block.py:57-136 with bf16 rounding at the same storage points, for every shape the kernel accepts, in inference
(no dropout, nothing saved) and training form (three dropout sites, all saved tensors checked).  Tolerance: 1e-2 of
the tensor's scale at the first convolution, widening to 2.5e-2 at the block output (six rounding points later)."""
import pytest
import torch
import torch.nn.functional as F

from dm_amd.runtime import hip, ops, pack
from oracle import ref_cpu as O
from oracle.noise import hash_normal, hash_uniform
from test_gpu_conv import assert_rel, nchw, nhwc, r16
from test_gpu_layers import bits

pytestmark = pytest.mark.gpu

SHAPES = [(2, 32, 16), (3, 64, 8), (5, 128, 4), (11, 256, 2), (2, 128, 8), (3, 256, 4), (16, 256, 2), (2, 32, 32), (3, 64, 16)]
FWD_ONLY_SHAPES = [(3, 256, 8), (2, 128, 16)]    # the two deepest stages of 128x128 images: forward in one launch, backward per unit


def make_params(c, e, seed):
    hn = lambda shape, i, s=1.0: hash_normal(shape, seed * 100 + i) * s          # noqa: E731
    return dict(we=hn((e, c, 1, 1), 1, (2.0 / c) ** 0.5), be=hn((e,), 2, 0.1), g1=1 + hn((e,), 3, 0.2), b1=hn((e,), 4, 0.2),
                wd=hn((e, 1, 3, 3), 5, 0.4), bd=hn((e,), 6, 0.1), g2=1 + hn((e,), 7, 0.2), b2=hn((e,), 8, 0.2),
                wp=hn((c, e, 1, 1), 9, (1.0 / e) ** 0.5), bp=hn((c,), 10, 0.1), g3=1 + hn((c,), 11, 0.2), b3=hn((c,), 12, 0.2))


def reference(x, P, act, keeps, scale):
    a = O.act_fn(act)
    drop = lambda v, k: v if k is None else v * k.permute(0, 3, 1, 2) * scale    # noqa: E731
    e_ch = P["wd"].shape[0]
    out = {}
    out["raw_e"] = r16(F.conv2d(r16(drop(x, keeps[0])), r16(P["we"]), P["be"]))
    out["y_e"] = r16(drop(a(F.group_norm(out["raw_e"], 16, P["g1"], P["b1"], eps=1e-5)), keeps[1]))
    out["raw_d"] = r16(F.conv2d(out["y_e"], P["wd"], P["bd"], padding=1, groups=e_ch))
    out["y_d"] = r16(drop(a(F.group_norm(out["raw_d"], 16, P["g2"], P["b2"], eps=1e-5)), keeps[2]))
    out["raw_o"] = r16(F.conv2d(out["y_d"], r16(P["wp"]), P["bp"]))
    out["y"] = r16(F.group_norm(out["raw_o"], 16, P["g3"], P["b3"], eps=1e-5) + x)
    return out


def run(dev, x, P, act, keeps, scale, record):
    d = lambda t: t.to(dev)                                                      # noqa: E731
    we, wp = pack.pack_fwd(hip.CONV_1X1, P["we"]).to(dev), pack.pack_fwd(hip.CONV_1X1, P["wp"]).to(dev)
    wd = P["wd"].reshape(-1, 9).t().contiguous().to(dev)
    masks = tuple(None if k is None else bits(k).to(dev) for k in keeps)
    return ops.sir_forward(nhwc(x, dev), we, d(P["be"]), d(P["g1"]), d(P["b1"]), wd, d(P["bd"]), d(P["g2"]), d(P["b2"]),
                           wp, d(P["bp"]), d(P["g3"]), d(P["b3"]), 16, hip.ACT[act], masks=masks, mask_scale=scale, record=record)


@pytest.mark.parametrize("n,c,w", SHAPES + FWD_ONLY_SHAPES)
@pytest.mark.parametrize("act", ["gelu", "silu"])
def test_fused_block_inference(dev, n, c, w, act):
    e = 2 * c
    assert ops.sir_supported(c, w, w, e, 16, hip.ACT[act])
    x = r16(hash_normal((n, c, w, w), 7) * 1.2)
    P = make_params(c, e, c + w)
    ref = reference(x, P, act, (None, None, None), 1.0)
    y, saved = run(dev, x, P, act, (None, None, None), 1.0, record=False)
    assert saved is None
    assert_rel(nchw(y), ref["y"], 2.5e-2)
    # the inference instantiation (no saved tensors, no masks: compiled without those paths) against the recording one
    y_rec, _ = run(dev, x, P, act, (None, None, None), 1.0, record=True)
    assert torch.equal(y, y_rec)


@pytest.mark.parametrize("n,c,w", SHAPES + FWD_ONLY_SHAPES)
def test_fused_block_training_saves_what_backward_needs(dev, n, c, w):
    e = 2 * c
    x = r16(hash_normal((n, c, w, w), 9) * 1.2)
    P = make_params(c, e, 3 * c + w)
    keeps = (hash_uniform((n, w, w, c), 21) < 0.9, hash_uniform((n, w, w, e), 22) < 0.9, hash_uniform((n, w, w, e), 23) < 0.9)
    ref = reference(x, P, "gelu", keeps, 1 / 0.9)
    y, sv = run(dev, x, P, "gelu", keeps, 1 / 0.9, record=True)
    for key, tol in (("raw_e", 1e-2), ("y_e", 1e-2), ("raw_d", 1.5e-2), ("y_d", 2e-2), ("raw_o", 2e-2)):
        assert_rel(nchw(sv[key]), ref[key], tol)
    assert_rel(nchw(y), ref["y"], 2.5e-2)
    for st, raw in (("st1", "raw_e"), ("st2", "raw_d"), ("st3", "raw_o")):
        got_raw = nchw(sv[raw])                       # statistics must describe the raw tensor the kernel stored
        g = got_raw.reshape(n, 16, -1)
        torch.testing.assert_close(sv[st][..., 0].cpu(), g.mean(-1), rtol=1e-3, atol=2e-3)
        torch.testing.assert_close(sv[st][..., 1].cpu(), (g.var(-1, unbiased=False) + 1e-5).rsqrt(), rtol=2e-3, atol=1e-3)


def test_unsupported_shapes_are_refused(dev):
    assert not ops.sir_supported(64, 32, 32, 128, 16, hip.ACT["gelu"])      # too large for one workgroup's LDS
    assert ops.sir_supported(32, 32, 32, 64, 16, hip.ACT["gelu"]) and ops.sir_backward_supported(32, 32, 32, 64, 16, hip.ACT["gelu"])
    assert not ops.sir_supported(64, 8, 8, 256, 16, hip.ACT["gelu"])        # expansion factor other than 2
    assert not ops.sir_supported(64, 8, 8, 128, 8, hip.ACT["gelu"])         # other group count


@pytest.mark.parametrize("n,c,w", SHAPES)
@pytest.mark.parametrize("train", [True, False])
def test_fused_block_backward_vs_autograd(dev, n, c, w, train):
    """One-launch data-gradient chain (+ per-channel parameter-gradient partials) against torch autograd through the
    restated block.  Data gradients pass 3-6 bf16 rounding points: 3e-2 of scale; parameter gradients (fp32 sums) 2e-2."""
    e = 2 * c
    act = "gelu" if train else "silu"
    x = r16(hash_normal((n, c, w, w), 31) * 1.2).requires_grad_(True)
    P = {k: v.requires_grad_(True) for k, v in make_params(c, e, 5 * c + w).items()}
    if train:
        keeps, scale = (hash_uniform((n, w, w, c), 41) < 0.9, hash_uniform((n, w, w, e), 42) < 0.9, hash_uniform((n, w, w, e), 43) < 0.9), 1 / 0.9
    else:
        keeps, scale = (None, None, None), 1.0
    ref = reference(x, P, act, keeps, scale)
    gy = r16(hash_normal((n, c, w, w), 33))
    names = ["we", "be", "g1", "b1", "wd", "bd", "g2", "b2", "wp", "bp", "g3", "b3"]
    grads = torch.autograd.grad(ref["y"], [x, ref["raw_o"], ref["raw_e"]] + [P[k] for k in names], gy)
    gx, g_o, g_e = grads[:3]
    gp = dict(zip(names, grads[3:]))

    Pd = {k: v.detach() for k, v in P.items()}
    y, sv = run(dev, x.detach(), Pd, act, keeps, scale, record=True)
    d = lambda t: t.detach().to(dev)                                             # noqa: E731
    wpd = pack.pack_dgrad(hip.CONV_1X1, Pd["wp"]).to(dev)
    wed = pack.pack_dgrad(hip.CONV_1X1, Pd["we"]).to(dev)
    wdf = Pd["wd"].reshape(-1, 9).flip(1).t().contiguous().to(dev)
    nwg = ops.sir_workgroups(w * w, n)
    nan = float("nan")
    p1, p2, p3 = (torch.full((n, 3, ch), nan, device=dev) for ch in (e, e, c))
    dwp = torch.full((nwg, 10, e), nan, device=dev)
    masks = tuple(None if k is None else bits(k).to(dev) for k in keeps)
    dx, d_o, d_e = ops.sir_backward(nhwc(gy, dev), sv, d(P["g1"]), d(P["b1"]), d(P["g2"]), d(P["b2"]), d(P["g3"]), wpd, wed, wdf, 16,
                                    hip.ACT[act], p1, p2, p3, dwp, masks=masks, mask_scale=scale)
    assert_rel(nchw(d_o), g_o, 1.5e-2)
    assert_rel(nchw(d_e), g_e, 3e-2)
    assert_rel(nchw(dx), gx, 3e-2)
    p1, p2, p3, dwp = p1.cpu(), p2.cpu(), p3.cpu(), dwp.cpu()
    assert_rel(p3[:, 0].sum(0), gp["g3"], 2e-2); assert_rel(p3[:, 1].sum(0), gp["b3"], 2e-2); assert_rel(p3[:, 2].sum(0), gp["bp"], 2e-2)
    assert_rel(p2[:, 0].sum(0), gp["g2"], 2e-2); assert_rel(p2[:, 1].sum(0), gp["b2"], 2e-2)
    assert_rel(p1[:, 0].sum(0), gp["g1"], 2e-2); assert_rel(p1[:, 1].sum(0), gp["b1"], 2e-2); assert_rel(p1[:, 2].sum(0), gp["be"], 2e-2)
    dw = dwp.sum(0)
    assert_rel(dw[:9].t().reshape(e, 1, 3, 3), gp["wd"], 2e-2)
    assert_rel(dw[9], gp["bd"], 2e-2)
