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
        SeparableInvertResidual(32, 64, expand_factor=2, activation="gelu")
