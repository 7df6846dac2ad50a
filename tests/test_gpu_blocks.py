"""GPU parity of the block-level modules (SeparableInvertResidual, DownBlock, UpBlock, EmbedBlock) against fixtures the
reference's own classes produced (tests/golden/blocks.pt, tools/gen_golden.py): forward output, input gradients and
every parameter gradient.  A block chains 6-30 bf16-storage kernels: outputs are held to 3e-2 of their scale, gradients
to a relative L2 error of 0.1 and cosine 0.99 per tensor (0.15 / 0.98 for the 30-kernel Down/Up blocks)."""
import pytest
import torch

from dm_amd.neural_nets.conv_block import SeparableInvertResidual
from dm_amd.neural_nets.denoise_models.constituent_blocks import DownBlock, EmbedBlock, SinusoidalPositionEmbedBlock, UpBlock

pytestmark = pytest.mark.gpu
KW = dict(expand_factor=2, drop_p=0.0, base_groups_norm=4, initializer="he_normal")


def l2(got, ref):
    return ((got - ref).norm() / (ref.norm() + 1e-12)).item()


def cos(a, b):
    return (a.flatten() @ b.flatten() / (a.norm() * b.norm() + 1e-20)).item()


def check(module, fx, inputs, ginputs, dev, l2_tol, cos_tol):
    assert list(module.state_dict().keys()) == list(fx["params"].keys())
    module.load_state_dict(fx["params"])
    module = module.to(dev).eval()
    xs = [fx[k].to(dev).requires_grad_(True) for k in inputs]
    y = module(*xs)
    ref = fx["y"]
    assert ((y.detach().cpu() - ref).abs().max() / ref.abs().max()).item() < 3e-2
    y.backward(fx["gy"].to(dev))
    for x, gk in zip(xs, ginputs):
        assert l2(x.grad.cpu(), fx[gk]) < l2_tol and cos(x.grad.cpu(), fx[gk]) > cos_tol, gk
    for name, p in module.named_parameters():
        g, r = p.grad.cpu(), fx["gparams"][name]
        assert l2(g, r) < l2_tol and cos(g, r) > cos_tol, (name, l2(g, r), cos(g, r))


@pytest.mark.parametrize("act", ["gelu", "silu"])
def test_separable_invert_residual(golden, dev, act):
    check(SeparableInvertResidual(32, 32, activation=act, **KW), golden("blocks.pt")[f"sir.{act}"], ["x"], ["gx"], dev, 0.1, 0.99)


@pytest.mark.parametrize("tag", ["c32to64.s1", "c32to64.s2", "c64.s2"])
def test_separable_invert_residual_without_residual_and_with_stride(golden, dev, tag):
    """the forms of the block the reference's UNet never builds (conv_block.py:51-54,84-86 there): another output width,
    stride 2 in the depthwise convolution; fixtures from the reference class"""
    fx = golden("sir_variants.pt")[tag]
    blk = SeparableInvertResidual(fx["in_channels"], fx["out_channels"], stride=fx["stride"], activation="gelu", **KW)
    assert not blk.is_residual
    check(blk, fx, ["x"], ["gx"], dev, 0.1, 0.99)


@pytest.mark.parametrize("act", ["gelu", "silu"])
def test_down_block(golden, dev, act):
    check(DownBlock(in_channels=32, out_channels=64, is_conditional=False, activation=act, **KW), golden("blocks.pt")[f"down.{act}"],
          ["x"], ["gx"], dev, 0.15, 0.98)


@pytest.mark.parametrize("act", ["gelu", "silu"])
def test_up_block(golden, dev, act):
    check(UpBlock(in_channels=64, out_channels=32, is_conditional=False, activation=act, **KW), golden("blocks.pt")[f"up.{act}"],
          ["x", "skip"], ["gx", "gskip"], dev, 0.15, 0.98)


@pytest.mark.parametrize("act", ["gelu", "silu"])
def test_embed_block(golden, dev, act):
    fx = golden("blocks.pt")[f"embed.{act}"]
    sin = SinusoidalPositionEmbedBlock(32)(fx["t"].to(dev).float() / 100)
    torch.testing.assert_close(sin.cpu(), fx["sin"], rtol=1e-5, atol=1e-6)
    m = EmbedBlock(32, 64, act, "he_normal")
    m.load_state_dict(fx["params"])
    m = m.to(dev)
    y = m(sin)
    assert y.shape == fx["y"].shape
    assert ((y.detach().cpu() - fx["y"]).abs().max() / fx["y"].abs().max()).item() < 1e-2
    y.sum().backward()                                     # gradients flow to all four parameters
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_train_mode_dropout_and_module_plumbing(dev):
    import copy
    m = SeparableInvertResidual(64, 64, expand_factor=2, drop_p=0.2, activation="gelu", initializer="he_normal").to(dev)
    x = torch.randn(4, 64, 8, 8, device=dev)
    m.train()
    a, b = m(x), m(x)
    assert not torch.equal(a, b)                           # fresh dropout masks per call
    m.eval()
    assert torch.equal(m(x), m(x))
    m2 = copy.deepcopy(m)
    assert torch.equal(m2(x), m(x)) and len(m.state_dict()) == 12
    with pytest.raises(ValueError):
        SeparableInvertResidual(32, 48, expand_factor=2, activation="gelu")          # widths are multiples of 32
    with pytest.raises(ValueError):
        SeparableInvertResidual(32, 32, expand_factor=2, stride=3, activation="gelu")


# ---- ConditionalModule (constituent_blocks.py:331-437 of the reference): fixtures from the reference class itself, for
# self-attention, cross-attention with the packed in_proj_weight and with separate q/k/v projection weights.
# A depth-1 module chains ~8 bf16-storage kernels per direction: output to 3e-2 of its scale, gradients to L2 0.1 / cosine 0.99.
@pytest.mark.parametrize("name", ["self_c128_s256", "self_c32_s256", "cross_packed", "cross_separate"])
def test_conditional_module(golden, dev, name):
    from dm_amd.neural_nets.denoise_models.constituent_blocks import ConditionalModule
    from oracle import ref_cpu as O
    from oracle.noise import hash_normal
    c = golden("conditional_module.pt")["cases"][name]
    mod = ConditionalModule(c["C"], c["activation"], c["depth"], 0.0, c["kdim"], c["kdim"])
    params = O.hash_conditional_params("", c["C"], c["depth"], c["kdim"], seed_base=c["param_seed_base"])
    assert list(mod.state_dict()) == list(params)
    mod.load_state_dict(params)
    mod = mod.to(dev).eval()
    q = hash_normal((2, c["S"], c["C"]), c["q_seed"]).to(dev).requires_grad_(True)
    cond = q if c["Sk"] is None else hash_normal((2, c["Sk"], c["kdim"]), c["cond_seed"]).to(dev).requires_grad_(True)
    y = mod(q, cond)
    assert y.shape == c["y"].shape
    assert ((y.detach().cpu() - c["y"]).abs().max() / c["y"].abs().max()).item() < 3e-2
    y.backward(hash_normal(tuple(y.shape), c["gy_seed"]).to(dev))
    assert l2(q.grad.cpu(), c["dq"]) < 0.1 and cos(q.grad.cpu(), c["dq"]) > 0.99, (l2(q.grad.cpu(), c["dq"]), cos(q.grad.cpu(), c["dq"]))
    if c["Sk"] is not None:
        assert l2(cond.grad.cpu(), c["dcond"]) < 0.1 and cos(cond.grad.cpu(), c["dcond"]) > 0.99
    for k, p in mod.named_parameters():
        g, r = p.grad.cpu(), c["dparams"][k]
        assert l2(g, r) < 0.1 and cos(g, r) > 0.99, (k, l2(g, r), cos(g, r))


def test_conditional_module_reference_fixture_forward(golden, dev):
    """the depth-2 self-attention forward of the original fixture"""
    from dm_amd.neural_nets.denoise_models.constituent_blocks import ConditionalModule
    from oracle import ref_cpu as O
    g = golden("conditional_module.pt")
    mod = ConditionalModule(g["C"], "gelu", g["depth"], 0.0, g["C"], g["C"])
    mod.load_state_dict(O.hash_conditional_params("", g["C"], g["depth"], None, seed_base=g["param_seed_base"]))
    mod = mod.to(dev).eval()
    q = g["q"].to(dev)
    with torch.no_grad():
        y = mod(q, q)
    assert ((y.cpu() - g["y_self"]).abs().max() / g["y_self"].abs().max()).item() < 3e-2


def test_conditional_module_depth2_self_gradients_vs_oracle(dev):
    """Two layers with condition = query: the second layer attends over the module's input (the reference hands the same
    condition to every layer), so the input collects gradients from both layers.  Checked against the oracle's autograd."""
    from dm_amd.neural_nets.denoise_models.constituent_blocks import ConditionalModule
    from oracle import ref_cpu as O
    from oracle.noise import hash_normal
    c, depth, s = 64, 2, 100
    params = O.hash_conditional_params("", c, depth, None, seed_base=1300)
    mod = ConditionalModule(c, "silu", depth, 0.0, c, c)
    mod.load_state_dict(params)
    mod = mod.to(dev).eval()
    q0, gy = hash_normal((3, s, c), 1301), hash_normal((3, s, c), 1302)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    qr = q0.clone().requires_grad_(True)
    yr = O.conditional_module(leaves, "", qr, qr, depth, O.act_fn("silu"))
    gr = torch.autograd.grad(yr, [qr] + list(leaves.values()), gy)
    q = q0.to(dev).requires_grad_(True)
    y = mod(q, q)
    assert ((y.detach().cpu() - yr.detach()).abs().max() / yr.abs().max()).item() < 3e-2
    y.backward(gy.to(dev))
    assert l2(q.grad.cpu(), gr[0]) < 0.1 and cos(q.grad.cpu(), gr[0]) > 0.99
    for (k, p), r in zip(mod.named_parameters(), gr[1:]):
        assert l2(p.grad.cpu(), r) < 0.1 and cos(p.grad.cpu(), r) > 0.99, (k, l2(p.grad.cpu(), r), cos(p.grad.cpu(), r))
