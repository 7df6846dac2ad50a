"""GPU parity of the autoencoder side of the latent-diffusion configuration (neural_nets/autoencoders/ of the reference):
BatchNorm2d + activation on the GroupNorm kernels against torch.nn.functional.batch_norm, then the Down/Up blocks and the
Encoder / Decoder stacks against fixtures the reference's own classes produced (tests/golden/autoencoder.pt,
tools/gen_golden.py): training-mode output, input and parameter gradients, the advanced running statistics, and the eval-mode
output.  Blocks chain 8-30 bf16-storage kernels: outputs are held to 3e-2 of their scale, gradients to a relative L2 error of
0.1 (0.15 for the stacks) per tensor."""
import pytest
import torch
import torch.nn.functional as F

from dm_amd.neural_nets.autoencoders import Decoder, DownBlock, Encoder, UpBlock
from dm_amd.runtime import hip, ops
from oracle import ref_cpu as O
from oracle.noise import hash_normal
from test_gpu_blocks import cos, l2
from test_gpu_conv import assert_rel, nchw, nhwc, r16

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,c,h,act", [(4, 64, 4, "gelu"), (2, 32, 8, "hardswish"), (8, 128, 2, "silu"), (3, 96, 6, "none"),
                                       (8, 32, 32, "gelu"), (2, 64, 64, "swish")])          # the last two: many-workgroup path
def test_batchnorm_act_training_and_eval(dev, n, c, h, act):
    x = r16(hash_normal((n, c, h, h), 1) * 1.5 + 0.6)
    gamma, beta = hash_normal((c,), 2) * 0.3 + 1, hash_normal((c,), 3) * 0.3
    rm, rv = hash_normal((c,), 4) * 0.2, hash_normal((c,), 5).abs() * 0.3 + 0.8
    gy = r16(hash_normal((n, c, h, h), 6))
    b = torch.tensor(1.2)
    scalar = act in ("swish", "hardswish")
    code = hip.ACT[act]

    def fa(z, bb):
        if act == "none":
            return z
        if scalar:
            return z * (torch.sigmoid(bb * z) if act == "swish" else F.hardsigmoid(bb * z))
        return O.act_fn(act)(z)

    for training in (True, False):
        xr, gr, br, bb = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True), b.clone().requires_grad_(True)
        rm_ref, rv_ref = rm.clone(), rv.clone()
        ref = fa(F.batch_norm(xr, rm_ref, rv_ref, gr, br, training, 0.1, 1e-5), bb)
        grads = torch.autograd.grad(ref, [xr, gr, br] + ([bb] if scalar else []), gy)
        rm_d, rv_d, nb = rm.to(dev), rv.to(dev), torch.zeros((), dtype=torch.long, device=dev)
        bd = b.to(dev) if scalar else None
        y, stats = ops.batchnorm_act_forward(nhwc(x, dev), gamma.to(dev), beta.to(dev), code, training=training, running_mean=rm_d,
                                             running_var=rv_d, num_batches=nb, act_beta=bd)
        assert_rel(nchw(y), ref.detach())
        torch.testing.assert_close(rm_d.cpu(), rm_ref, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(rv_d.cpu(), rv_ref, rtol=1e-3, atol=1e-5)
        assert int(nb) == (1 if training else 0)
        part = torch.full((3, c), float("nan"), device=dev)
        dpart = torch.full((c,), float("nan"), device=dev) if scalar else None
        dx = ops.batchnorm_act_backward(nhwc(x, dev), nhwc(gy, dev), gamma.to(dev), beta.to(dev), code, stats, part, training=training,
                                        act_beta=bd, dbeta_part=dpart)
        assert_rel(nchw(dx), grads[0], 1.5e-2)
        assert_rel(part[0].cpu(), grads[1], 1e-2)
        assert_rel(part[1].cpu(), grads[2], 1e-2)
        if scalar:
            assert abs(dpart.sum().item() - grads[3].item()) < 1e-2 * max(1.0, abs(grads[3].item()))


def _load(mod, rec, dev):
    assert list(mod.state_dict()) == list(rec["shapes"])
    mod.load_state_dict(O.hash_state(rec["shapes"], rec["seed"]))
    return mod.to(dev)


def _check(mod, rec, dev, l2_tol):
    mod = _load(mod, rec, dev).train()
    x = rec["x"].to(dev).requires_grad_(True)
    y = mod(x)
    ref = rec["y"]
    assert ((y.detach().cpu() - ref).abs().max() / ref.abs().max()).item() < 3e-2
    y.backward(hash_normal(tuple(ref.shape), rec["gy_seed"]).to(dev))
    assert l2(x.grad.cpu(), rec["gx"]) < l2_tol and cos(x.grad.cpu(), rec["gx"]) > 0.98
    typical = sorted(d["l2"] for d in rec["gdigest"].values())[len(rec["gdigest"]) // 2]
    for name, p in mod.named_parameters():
        d, full = rec["gdigest"][name], rec["gparams"][name]
        assert p.grad is not None, name
        if name.endswith(".beta") or p.numel() == 1:
            # an activation scalar's gradient is a signed sum over the whole layer whose terms switch at the hard gate's
            # corners: held to the typical gradient scale (the GroupNorm-level tests pin the arithmetic)
            assert abs(p.grad.item() - full.item()) < 0.5 * max(typical, abs(full.item())), (name, p.grad.item(), full.item())
            continue
        if d["l2"] < 1e-3 * typical:
            continue                     # analytically zero (a bias in front of a norm): round-off in the reference
        assert abs(p.grad.double().norm().item() - d["l2"]) < l2_tol * d["l2"], (name, p.grad.norm().item(), d["l2"])
        if full is not None:
            assert l2(p.grad.cpu(), full) < l2_tol and cos(p.grad.cpu(), full) > 0.98, (name, l2(p.grad.cpu(), full))
    for k, v in rec["buffers_after"].items():            # the running statistics advanced as the reference's did
        got = dict(mod.named_buffers())[k].cpu()
        if v.dtype.is_floating_point:
            assert (got - v).abs().max().item() < 2e-2 * max(1.0, v.abs().max().item()), k
        else:
            assert int(got) == int(v) == 1
    mod.eval()
    with torch.no_grad():
        ye = mod(rec["x"].to(dev))
    assert ((ye.cpu() - rec["y_eval"]).abs().max() / rec["y_eval"].abs().max()).item() < 3e-2
    assert all(int(b) == 1 for k, b in mod.named_buffers() if k.endswith("num_batches_tracked"))     # eval leaves the statistics alone


@pytest.mark.parametrize("act", ["gelu", "hardswish"])
def test_autoencoder_down_up_blocks(golden, dev, act):
    g = golden("autoencoder.pt")
    kw = dict(expand_factor=2, drop_p=0.0, activation=act, initializer="he_uniform")
    _check(DownBlock(32, 64, **kw), g[f"down.{act}"], dev, 0.1)
    _check(UpBlock(64, 32, **kw), g[f"up.{act}"], dev, 0.1)


def test_autoencoder_encoder_decoder(golden, dev):
    g = golden("autoencoder.pt")
    kw = dict(expand_factor=2, drop_p=0.0, activation="hardswish", initializer="he_uniform")
    _check(Encoder(3, (64, 128), 128, **kw), g["encoder.hardswish"], dev, 0.15)
    _check(Decoder(3, 128, (128, 64), **kw), g["decoder.hardswish"], dev, 0.15)


def test_autoencoder_round_trip_trains(dev):
    """Encoder -> Decoder on 64x64 RGB with dropout on, one optimizer step through plain autograd over the two modules:
    shapes, finiteness, and every parameter of both modules receives a gradient"""
    kw = dict(expand_factor=2, drop_p=0.1, activation="hardswish")
    enc, dec = Encoder(3, (64, 128), 128, **kw).to(dev), Decoder(3, 128, (128, 64), **kw).to(dev)
    x = (hash_normal((4, 3, 64, 64), 9) * 0.5).to(dev)
    z = enc(x)
    assert z.shape == (4, 128, 16, 16)
    out = dec(z)
    assert out.shape == (4, 3, 128, 128)               # the decoder's first transposed convolution doubles once more than the encoder halves
    loss = out.square().mean()
    loss.backward()
    for m in (enc, dec):
        for k, p in m.named_parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
    assert torch.isfinite(loss)
