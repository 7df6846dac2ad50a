"""GPU parity of the attention core (forward, dQ/dK/dV, dropout consistency) and of the per-sample normalisation
kernels, each against torch fp32 on the CPU with identical bf16-rounded operands.  Tolerance: 1e-2 of the tensor's
scale (bf16 operands and probabilities, fp32 accumulation), 1.5e-2 for gradients."""
import math

import pytest
import torch
import torch.nn.functional as F

from dm_amd.runtime import hip, ops
from oracle import ref_cpu as O
from oracle.noise import hash_normal
from test_gpu_conv import assert_rel, r16

pytestmark = pytest.mark.gpu


def ref_attention(q, k, v, heads, keep=None, scale=1.0):
    n, s, c = q.shape
    sk, dh = k.shape[1], c // heads
    qh, kh, vh = (t.view(n, -1, heads, dh).transpose(1, 2) for t in (q, k, v))
    p = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(dh), dim=-1)
    if keep is not None:
        p = p * keep * scale
    return (p @ vh).transpose(1, 2).reshape(n, s, c)


def tok(t, dev):
    return t.to(dev).bfloat16().unsqueeze(2).contiguous()


# (N, S, Sk, C, heads): head dims 4, 8, 16, 32, 64; ragged token counts; cross-attention lengths
CASES = [(2, 256, 256, 128, 8), (2, 64, 64, 32, 8), (1, 64, 64, 64, 8), (2, 16, 16, 256, 8), (1, 4, 4, 512, 8), (2, 100, 77, 128, 8),
         (1, 300, 40, 64, 2), (1, 1024, 1024, 128, 8)]


@pytest.mark.parametrize("n,s,sk,c,heads", CASES)
def test_attention_forward_backward(dev, n, s, sk, c, heads):
    q, k, v = (r16(hash_normal((n, t, c), 10 + i) * (1.5 if i < 2 else 1.0)) for i, t in enumerate((s, sk, sk)))
    do = r16(hash_normal((n, s, c), 20))
    ql, kl, vl = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref = ref_attention(ql, kl, vl, heads)
    gq, gk, gv = torch.autograd.grad(ref, [ql, kl, vl], do)

    o, lse = ops.attention_forward(tok(q, dev), tok(k, dev), tok(v, dev), heads)
    assert_rel(o.float().cpu().view(n, s, c), ref.detach())
    dq, dk, dv = (torch.full((n, t, 1, c), float("nan"), device=dev, dtype=torch.bfloat16) for t in (s, sk, sk))
    ops.attention_backward(tok(q, dev), tok(k, dev), tok(v, dev), o, lse, tok(do, dev), dq, dk, dv, heads)
    assert_rel(dq.float().cpu().view(n, s, c), gq, 1.5e-2)
    assert_rel(dk.float().cpu().view(n, sk, c), gk, 1.5e-2)
    assert_rel(dv.float().cpu().view(n, sk, c), gv, 1.5e-2)


def test_attention_packed_projection_views(dev):
    """q, k, v as channel slices of one [N,S,1,3C] buffer (the layout the packed in_proj GEMM produces); gradients
    written into slices of one packed buffer."""
    n, s, c, heads = 2, 64, 128, 8
    qkv = r16(hash_normal((n, s, 3 * c), 31))
    do = r16(hash_normal((n, s, c), 32))
    leaf = qkv.clone().requires_grad_(True)
    ref = ref_attention(leaf[..., :c], leaf[..., c:2 * c], leaf[..., 2 * c:], heads)
    g, = torch.autograd.grad(ref, [leaf], do)
    buf = tok(qkv, dev)
    qd, kd, vd = buf[..., :c], buf[..., c:2 * c], buf[..., 2 * c:]
    o, lse = ops.attention_forward(qd, kd, vd, heads)
    assert_rel(o.float().cpu().view(n, s, c), ref.detach())
    dbuf = torch.full_like(buf, float("nan"))
    ops.attention_backward(qd, kd, vd, o, lse, tok(do, dev), dbuf[..., :c], dbuf[..., c:2 * c], dbuf[..., 2 * c:], heads)
    assert_rel(dbuf.float().cpu().view(n, s, 3 * c), g, 1.5e-2)


def test_attention_dropout_is_consistent(dev):
    """With v = identity-like one-hot rows the forward output exposes dropout(P) itself: recover the keep mask from it,
    check its rate and scale, and check that backward used the same mask (gradients match torch with that mask)."""
    n, s, heads, dh = 1, 64, 2, 64
    c = heads * dh
    q, k = (r16(hash_normal((n, s, c), 40 + i) * 0.3) for i in range(2))
    v = torch.eye(s).repeat(1, heads).view(1, s, c).contiguous()            # head h: v[key] = e_key  ->  o[q, h, :] = dropout(P)[q, :]
    seed = torch.tensor([1234567], device=dev, dtype=torch.long)
    p_drop = 0.25
    o, lse = ops.attention_forward(tok(q, dev), tok(k, dev), tok(v, dev), heads, drop_p=p_drop, seed=seed, stream_id=3)
    o2, _ = ops.attention_forward(tok(q, dev), tok(k, dev), tok(v, dev), heads, drop_p=p_drop, seed=seed, stream_id=3)
    assert torch.equal(o, o2)
    pd = o.float().cpu().view(n, s, heads, s).transpose(1, 2)               # [n, heads, q, key]
    keep = (pd != 0).float()
    rate = keep.mean().item()
    assert abs(rate - 0.75) < 0.02, rate
    ql, kl, vl = (t.clone().requires_grad_(True) for t in (q, k, v))
    scale = 256.0 / round(0.75 * 256)
    ref = ref_attention(ql, kl, vl, heads, keep=keep, scale=scale)
    assert_rel(o.float().cpu().view(n, s, c), ref.detach(), 2e-2)
    do = r16(hash_normal((n, s, c), 45))
    gq, gk, gv = torch.autograd.grad(ref, [ql, kl, vl], do)
    dq, dk, dv = (torch.empty(n, s, 1, c, device=dev, dtype=torch.bfloat16) for _ in range(3))
    ops.attention_backward(tok(q, dev), tok(k, dev), tok(v, dev), o, lse, tok(do, dev), dq, dk, dv, heads, drop_p=p_drop, seed=seed, stream_id=3)
    assert_rel(dq.float().cpu().view(n, s, c), gq, 2e-2)
    assert_rel(dk.float().cpu().view(n, s, c), gk, 2e-2)
    assert_rel(dv.float().cpu().view(n, s, c), gv, 2e-2)


@pytest.mark.parametrize("n,s,c,act,with_b", [(2, 256, 128, "none", True), (3, 64, 32, "gelu", True), (2, 16, 512, "silu", True),
                                               (2, 100, 64, "gelu", False), (1, 1024, 256, "gelu", True)])
def test_sample_norm_fwd_bwd(dev, n, s, c, act, with_b):
    a = r16(hash_normal((n, s, c), 50) * 1.3 + 0.2)
    b = r16(hash_normal((n, s, c), 51)) if with_b else None
    gamma, beta = hash_normal((c,), 52) * 0.3 + 1, hash_normal((c,), 53) * 0.3
    dy = r16(hash_normal((n, s, c), 54))
    f = O.act_fn(act) if act != "none" else (lambda t: t)
    al = a.clone().requires_grad_(True)
    bl = b.clone().requires_grad_(True) if with_b else None
    gl, be = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    u = f(al) + (bl if with_b else 0)
    ref = F.group_norm(u.transpose(1, 2).unsqueeze(-1), 1, gl, be, eps=1e-5).squeeze(-1).transpose(1, 2)
    leaves = [al, gl, be] + ([bl] if with_b else [])
    grads = torch.autograd.grad(ref, leaves, dy)

    ad, bd = tok(a, dev), (tok(b, dev) if with_b else None)
    y, stats = ops.sample_norm_forward(ad, bd, hip.ACT[act], gamma.to(dev), beta.to(dev))
    assert_rel(y.float().cpu().view(n, s, c), ref.detach())
    part = torch.full((n, 3, c), float("nan"), device=dev)
    du, da = ops.sample_norm_backward(ad, bd, tok(dy, dev), hip.ACT[act], gamma.to(dev), stats, part)
    assert_rel(da.float().cpu().view(n, s, c), grads[0], 1.5e-2)
    if with_b:
        assert_rel(du.float().cpu().view(n, s, c), grads[3], 1.5e-2)
    ps = part.sum(0).cpu()
    assert_rel(ps[0], grads[1], 1e-2)
    assert_rel(ps[1], grads[2], 1e-2)
    assert_rel(ps[2], grads[0].sum((0, 1)), 2e-2)
