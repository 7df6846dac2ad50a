"""GPU parity of the all-MLPs-in-one-launch time-embedding kernels against the oracle's EmbedBlock restatement
(constituent_blocks.py:282-328) and torch autograd, with bf16 rounding at the same storage points.  1e-2 of scale."""
import pytest
import torch
import torch.nn.functional as F

from dm_amd.runtime import hip, ops, pack
from oracle import ref_cpu as O
from oracle.noise import hash_normal
from test_gpu_conv import assert_rel, r16

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,act", [(40, "gelu"), (256, "silu")])
def test_all_embed_mlps_forward_backward(dev, n, act):
    widths = [32, 32, 64, 128, 256, 128, 64, 32]
    a = O.act_fn(act)
    sin = r16(O.sinusoidal(torch.randint(1, 121, (n,)).float() / 100.0, 32))
    params, refs = [], []
    for i, c in enumerate(widths):
        w0 = (hash_normal((2 * c, 32), 10 + i) * 0.25).requires_grad_(True)
        b0 = (hash_normal((2 * c,), 30 + i) * 0.1).requires_grad_(True)
        w1 = (hash_normal((c, 2 * c), 50 + i) * (1.0 / (2 * c)) ** 0.5).requires_grad_(True)
        b1 = (hash_normal((c,), 70 + i) * 0.1).requires_grad_(True)
        raw0 = r16(F.linear(sin, r16(w0), b0))
        h = r16(a(raw0))
        out = F.linear(h, r16(w1), b1)
        params.append((w0, b0, w1, b1))
        refs.append((raw0, h, out))
    d = lambda t: t.detach().to(dev)                                            # noqa: E731
    sin_d = sin.view(n, 1, 1, 32).to(dev).bfloat16()
    w0s = [pack.pack_fwd(hip.CONV_1X1, p[0].detach()[:, :, None, None]).to(dev) for p in params]
    w1s = [pack.pack_fwd(hip.CONV_1X1, p[2].detach()[:, :, None, None]).to(dev) for p in params]
    outs, saved = ops.embed_forward(sin_d, w0s, [d(p[1]) for p in params], w1s, [d(p[3]) for p in params], hip.ACT[act], record=True)
    for (raw0, h, out), o, sv in zip(refs, outs, saved):
        assert_rel(o.cpu(), out.detach())
        assert_rel(sv[0].float().cpu().view(n, -1), raw0.detach())
        assert_rel(sv[1].float().cpu().view(n, -1), h.detach())
    outs2, saved2 = ops.embed_forward(sin_d, w0s, [d(p[1]) for p in params], w1s, [d(p[3]) for p in params], hip.ACT[act], record=False)
    assert saved2 is None and all(torch.equal(x, y) for x, y in zip(outs, outs2))

    # backward data path + bias partials; the two weight gradients per MLP go through the dense weight-gradient kernel
    gouts = [hash_normal((n, c), 90 + i) for i, c in enumerate(widths)]
    tiles = (n + 31) // 32
    db1s = [torch.full((tiles, c), float("nan"), device=dev) for c in widths]
    db0s = [torch.full((tiles, 2 * c), float("nan"), device=dev) for c in widths]
    w1ds = [pack.pack_dgrad(hip.CONV_1X1, p[2].detach()[:, :, None, None]).to(dev) for p in params]
    dt16s, dzs = ops.embed_backward([d(g) for g in gouts], [sv[0] for sv in saved], w1ds, hip.ACT[act], db1s, db0s)
    for i, ((w0, b0, w1, b1), (raw0, h, out), g) in enumerate(zip(params, refs, gouts)):
        g_raw0, g_b0, g_b1, g_w0, g_w1 = torch.autograd.grad(out, [raw0, b0, b1, w0, w1], r16(g))
        assert_rel(dt16s[i].float().cpu().view(n, -1), r16(g), 1e-6)
        assert_rel(dzs[i].float().cpu().view(n, -1), g_raw0, 1.5e-2)
        assert_rel(db1s[i].sum(0).cpu(), g_b1)
        assert_rel(db0s[i].sum(0).cpu(), g_b0, 1.5e-2)
        c = widths[i]
        dw1 = torch.zeros(1, c, 2 * c, device=dev)
        ops.conv_wgrad(hip.CONV_1X1, saved[i][1], dt16s[i], c, dw1)
        assert_rel(dw1[0].cpu(), g_w1, 1.5e-2)
        dw0 = torch.zeros(1, 2 * c, 32, device=dev)
        ops.conv_wgrad(hip.CONV_1X1, sin_d, dzs[i], 2 * c, dw0)
        assert_rel(dw0[0].cpu(), g_w0, 1.5e-2)
