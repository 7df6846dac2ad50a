"""Pin the CPU oracle (oracle/ref_cpu.py) to fixtures produced by the reference itself
(tools/gen_golden.py).  CPU only; tolerances are fp32 round-off of re-associated sums."""
import math

import pytest
import torch

from oracle import ref_cpu as O
from oracle.noise import hash_normal, hash_uniform

RT, AT = 2e-5, 2e-6


def close(a, b, rtol=RT, atol=AT):
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol, equal_nan=True)


def test_schedule_tables_bit_exact(golden):
    g = golden("schedule_T120_b025.pt")
    s = O.make_schedule(g["T"], g["beta"])
    for k, v in g["tables"].items():
        assert s[k].shape == v.shape == (121, 1, 1, 1)
        assert torch.equal(torch.nan_to_num(s[k], nan=-7.0), torch.nan_to_num(v, nan=-7.0)), k
    # index-0 singularities of the reference tables (beta_0 = 0)
    assert math.isinf(s["x_previous_coef"][0].item()) and math.isnan(s["x0_pred_coef"][0].item())


def test_q_reverse_elbo(golden):
    g = golden("diffusion_ops.pt")
    s = O.make_schedule(120, 0.25)
    assert torch.equal(O.q_sample(s, g["x0"], g["t"], g["eps"]), g["x_t"])
    assert torch.equal(O.reverse_step(s, g["x_t"], g["x0_pred"], g["t"], g["z"]), g["x_prev"])
    # t = 1 returns the prediction itself (up to fp32 rounding of beta_1/(1-abar_1)), no noise
    close(g["x_prev"][0], g["x0_pred"][0], 5e-5, 5e-5)
    xp = g["x0_pred"].clone().requires_grad_(True)
    loss = O.elbo_loss(s, g["t"], xp, g["x0"], "mse")
    close(loss, g["loss_mse"])
    close(torch.autograd.grad(loss, xp)[0], g["grad_mse"])
    close(O.elbo_loss(s, g["t"], g["x0_pred"], (g["x0"] + 1) / 2, "bce"), g["loss_bce"])
    close(O.elbo_loss(s, g["t"], g["x0_pred"], g["x0"], "mse", use_elbo=False), g["loss_plain"])
    close(g["loss_mse_ddpflag"], g["loss_mse"])


@pytest.mark.parametrize("act", ["gelu", "silu"])
def test_blocks(golden, act):
    g = golden("blocks.pt")
    a = O.act_fn(act)

    def run(fn, rec, inputs):
        leaves = {k: v.clone().requires_grad_(True) for k, v in rec["params"].items()}
        xs = [rec[k].clone().requires_grad_(True) for k in inputs]
        y = fn(leaves, *xs)
        close(y, rec["y"], 1e-4, 1e-5)
        grads = torch.autograd.grad(y, xs + list(leaves.values()), rec["gy"])
        for k, gx in zip(inputs, grads):
            close(gx, rec["g" + k], 1e-4, 1e-4)
        for k, gp in zip(leaves, grads[len(xs):]):
            close(gp, rec["gparams"][k], 1e-4, 2e-4)

    run(lambda p, x: O.sir_block(p, "", x, 16, a, O.DropMasks()), g[f"sir.{act}"], ["x"])
    run(lambda p, x: O.down_block(p, "", x, 16, a, O.DropMasks()), g[f"down.{act}"], ["x"])
    run(lambda p, x, skip: O.up_block(p, "", x, skip, 16, a, O.DropMasks()), g[f"up.{act}"], ["x", "skip"])
    e = g[f"embed.{act}"]
    sin = O.sinusoidal(e["t"].float() / 100, 32)
    close(sin, e["sin"])
    close(O.embed_block(e["params"], "", sin, a), e["y"])


def test_inverted_residual_without_residual_and_with_stride(golden):
    """the block forms the reference's UNet never builds: other output width, stride 2 (conv_block.py:51-54,84-86)"""
    g = golden("sir_variants.pt")
    a = O.act_fn("gelu")
    for tag in ("c32to64.s1", "c32to64.s2", "c64.s2"):
        rec = g[tag]
        leaves = {k: v.clone().requires_grad_(True) for k, v in rec["params"].items()}
        x = rec["x"].clone().requires_grad_(True)
        y = O.sir_block(leaves, "", x, 16, a, O.DropMasks(), stride=rec["stride"])
        close(y, rec["y"], 1e-4, 1e-5)
        grads = torch.autograd.grad(y, [x] + list(leaves.values()), rec["gy"])
        close(grads[0], rec["gx"], 1e-4, 1e-4)
        for k, gp in zip(leaves, grads[1:]):
            close(gp, rec["gparams"][k], 1e-4, 2e-4)


def _digest_close(g, d, rtol):
    flat = g.reshape(-1).double()
    l2 = flat.norm().item()
    assert abs(l2 - d["l2"]) <= rtol * max(d["l2"], 1e-6), (l2, d["l2"])
    ref = d["val"].double()
    got = flat[d["idx"]]
    assert (got - ref).abs().max().item() <= rtol * max(d["l2"] / math.sqrt(flat.numel()), 1e-7) * 10 + rtol * ref.abs().max().item()


@pytest.mark.parametrize("tag,c,act", [("c3.gelu", 3, "gelu"), ("c1.gelu", 1, "gelu"), ("c3.silu", 3, "silu")])
def test_unet_full_forward_backward(golden, tag, c, act):
    g = golden("unet_full.pt")[tag]
    cfg = O.UNetCfg(img_channels=c, activation=act, drop_p=0.0)
    p = O.hash_params(cfg, seed=1)
    assert sum(v.numel() for v in p.values()) == g["n_params"] == (6857763 if c == 3 else 6856609)
    assert len(p) == 350
    s = O.make_schedule(120, 0.25)
    x_t = O.q_sample(s, g["x0"], g["t"], hash_normal(tuple(g["x0"].shape), 401))
    assert torch.equal(x_t, g["x_t"])
    taps = {}
    with torch.no_grad():
        pred = O.unet_forward(p, cfg, x_t, g["t"], taps=taps)
    for k, v in g["taps"].items():
        close(taps[k], v, 2e-4, 2e-5)
    close(pred, g["pred"], 2e-4, 2e-5)
    loss, grads = O.train_step_grads(p, cfg, s, g["x0"], g["t"], hash_normal(tuple(g["x0"].shape), 401))
    close(loss, g["loss"], 1e-4, 1e-7)
    for k, d in g["grad_digest"].items():
        _digest_close(grads[k], d, 2e-3)


def test_unet_dropout_sites(golden):
    g = golden("unet_full.pt")["c3.gelu.dropout"]
    cfg = O.UNetCfg(img_channels=3, activation="gelu", drop_p=0.1)
    assert g["sites"] == O.count_dropout_sites(cfg) == 60
    p = O.hash_params(cfg, seed=1)
    s = O.make_schedule(120, 0.25)
    x_t = O.q_sample(s, g["x0"], g["t"], hash_normal(tuple(g["x0"].shape), 401))
    shapes = []

    class Rec(O.DropMasks):
        def __call__(self, v):
            shapes.append(tuple(v.shape))
            return v

    with torch.no_grad():
        O.unet_forward(p, cfg, x_t, g["t"], drop=Rec())
    masks = [(hash_uniform(sh, 9000 + i) < 0.9).float() / 0.9 for i, sh in enumerate(shapes)]
    with torch.no_grad():
        pred = O.unet_forward(p, cfg, x_t, g["t"], drop=O.DropMasks(masks))
    close(pred, g["pred"], 2e-4, 2e-5)
    loss, grads = O.train_step_grads(p, cfg, s, g["x0"], g["t"], hash_normal(tuple(g["x0"].shape), 401), masks=masks)
    close(loss, g["loss"], 1e-4, 1e-7)
    for k in list(g["grad_digest"])[::25]:
        _digest_close(grads[k], g["grad_digest"][k], 2e-3)


def test_sample_chain(golden):
    g = golden("sample_chain.pt")
    cfg = O.UNetCfg(img_channels=3, activation="gelu", drop_p=0.0)
    p = O.hash_params(cfg, seed=1, head_gain=g["head_gain"])
    s = O.make_schedule(120, 0.25)
    shp = (g["N"], 3) + tuple(g["shape"])
    base = g["noise_seed_base"]
    x_T = hash_normal(shp, base + 1)
    noises = [hash_normal(shp, base + 2 + k) for k in range(120)]
    for k in (1, 2):
        close(O.sample_chain(p, cfg, s, x_T, 120, noises, steps=k), g["snapshots"][k], 2e-4, 2e-5)
    final = O.sample_chain(p, cfg, s, x_T, 120, noises)
    close(final, g["snapshots"][120], 1e-3, 1e-4)
    close(O.to_unit_range(final), g["final_unit"], 1e-3, 1e-4)


def test_conditional_module(golden):
    g = golden("conditional_module.pt")
    p = {}
    for i, (k, shp) in enumerate(g["shapes"].items()):
        if len(shp) >= 2:
            p[k] = O.hash_fill(shp, g["param_seed_base"] + i, (3.0 / shp[-1]) ** 0.5)
        elif "norm" in k and k.endswith("weight"):
            p[k] = O.hash_fill(shp, g["param_seed_base"] + i, 0.2, 1.0)
        else:
            p[k] = O.hash_fill(shp, g["param_seed_base"] + i, 0.1)
    y = O.conditional_module(p, "", g["q"], g["q"], g["depth"], O.act_fn("gelu"), g["heads"])
    close(y, g["y_self"], 2e-4, 2e-5)
    assert list(O.conditional_shapes("", g["C"], g["depth"]).keys()) == list(g["shapes"].keys())


@pytest.mark.parametrize("name", ["self_c128_s256", "self_c32_s256", "cross_packed", "cross_separate"])
def test_conditional_module_gradients(golden, name):
    """forward and autograd gradients of the restated ConditionalModule against the reference's, for self-attention and
    for cross-attention with packed and with separate projection weights"""
    c = golden("conditional_module.pt")["cases"][name]
    p = O.hash_conditional_params("", c["C"], c["depth"], c["kdim"], seed_base=c["param_seed_base"])
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    q = hash_normal((2, c["S"], c["C"]), c["q_seed"]).requires_grad_(True)
    cond = q if c["Sk"] is None else hash_normal((2, c["Sk"], c["kdim"]), c["cond_seed"]).requires_grad_(True)
    y = O.conditional_module(leaves, "", q, cond, c["depth"], O.act_fn(c["activation"]))
    close(y.detach(), c["y"], 2e-4, 2e-5)
    ins = [q] + ([] if c["Sk"] is None else [cond])
    grads = torch.autograd.grad(y, ins + list(leaves.values()), hash_normal(tuple(y.shape), c["gy_seed"]))
    close(grads[0], c["dq"], 1e-3, 1e-4)
    if c["Sk"] is not None:
        close(grads[1], c["dcond"], 1e-3, 1e-4)
    for k, gk in zip(leaves, grads[len(ins):]):
        close(gk, c["dparams"][k], 1e-3, 2e-4)


def test_philox_oracle_known_answers():
    """Random123 kat_vectors, philox4x32 with 10 rounds: (counter, key) -> output."""
    import numpy as np
    from oracle.philox import philox4x32_10, sample_timesteps
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = tuple(int(x[0]) for x in philox4x32_10(*[np.array([v]) for v in ctr], *key))
        assert got == want
    t = sample_timesteps(10000, 120, 42, 3)
    assert t.min() == 1 and t.max() == 120 and t.dtype == np.int64


@pytest.mark.parametrize("act", ["swish", "hardswish"])
def test_learnable_scalar_activations(golden, act):
    """swish / hardswish (neural_nets/activations.py:25-63 of the reference): one learnable scalar per activation module.
    The oracle against the reference's inverted-residual block, EmbedBlock, ConditionalModule and a small whole network,
    forward and gradients, the scalars' gradients included."""
    g = golden("beta_activations.pt")
    rec = g[f"sir.{act}"]
    leaves = {k: v.clone().requires_grad_(True) for k, v in rec["params"].items()}
    x = rec["x"].clone().requires_grad_(True)
    y = O.sir_block(leaves, "", x, 16, O.act_fn(act, leaves), O.DropMasks())
    close(y, rec["y"], 1e-4, 1e-5)
    grads = torch.autograd.grad(y, [x] + list(leaves.values()), rec["gy"])
    close(grads[0], rec["gx"], 1e-4, 1e-4)
    for k, gp in zip(leaves, grads[1:]):
        close(gp, rec["gparams"][k], 1e-4, 2e-4)
    assert rec["gparams"]["expansion.3.beta"].shape == () and abs(float(rec["gparams"]["expansion.3.beta"])) > 1.0

    e = g[f"embed.{act}"]
    leaves = {k: v.clone().requires_grad_(True) for k, v in e["params"].items()}
    ye = O.embed_block(leaves, "", e["sin"], O.act_fn(act, leaves))
    close(ye, e["y"])
    for k, gp in zip(leaves, torch.autograd.grad(ye, list(leaves.values()), e["gy"])):
        close(gp, e["gparams"][k], 1e-4, 1e-5)

    c = g[f"conditional.{act}"]
    p = {k: v.requires_grad_(True) for k, v in O.hash_conditional_params("", c["C"], 1, c["C"], seed_base=c["param_seed_base"], beta=True).items()}
    q = hash_normal((2, c["S"], c["C"]), c["q_seed"]).requires_grad_(True)
    yc = O.conditional_module(p, "", q, q, 1, O.act_fn(act, p))
    close(yc, c["y"], 1e-4, 1e-5)
    gc = torch.autograd.grad(yc, [q] + list(p.values()), hash_normal(tuple(yc.shape), c["gy_seed"]))
    close(gc[0], c["dq"], 1e-3, 1e-4)
    for k, gp in zip(p, gc[1:]):
        close(gp, c["dparams"][k], 1e-3, 2e-4)

    u = g[f"unet.{act}"]
    cfg = O.UNetCfg(img_channels=1, stages_channels=(32, 64), latent_dim=64, activation=act, drop_p=0.0)
    pu = O.hash_params(cfg, seed=1)
    assert sum(v.numel() for v in pu.values()) == u["n_params"]
    s = O.make_schedule(120, 0.25)
    noise = hash_normal(tuple(u["x0"].shape), 401)
    assert torch.equal(O.q_sample(s, u["x0"], u["t"], noise), u["x_t"])
    with torch.no_grad():
        close(O.unet_forward(pu, cfg, u["x_t"], u["t"]), u["pred"], 2e-4, 2e-5)
    loss, grads = O.train_step_grads(pu, cfg, s, u["x0"], u["t"], noise)
    close(loss, u["loss"], 1e-4, 1e-7)
    for k, d in u["grad_digest"].items():
        _digest_close(grads[k], d, 2e-3)
    for k, v in u["beta_grads"].items():
        close(grads[k], v, 2e-3, 1e-6)


def _ae_run(kind, rec, act_name, training=True):
    p = O.hash_state(rec["shapes"], rec["seed"])
    leaves = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in p.items()}
    act = O.act_fn(act_name, leaves)
    x = rec["x"].clone().requires_grad_(True)
    fn = {"down": lambda: O.ae_down_block(leaves, "", x, act, O.DropMasks(), training),
          "up": lambda: O.ae_up_block(leaves, "", x, act, O.DropMasks(), training),
          "encoder": lambda: O.ae_encoder(leaves, x, 2, act, O.DropMasks(), training),
          "decoder": lambda: O.ae_decoder(leaves, x, 2, act, O.DropMasks(), training)}[kind]
    return fn(), x, leaves


@pytest.mark.parametrize("tag", ["down.gelu", "down.hardswish", "up.gelu", "up.hardswish", "encoder.hardswish", "decoder.hardswish"])
def test_autoencoder_blocks(golden, tag):
    """The autoencoder Down/Up blocks and Encoder / Decoder stacks of the reference (bias-free convolutions + BatchNorm2d +
    activation around an inverted-residual block): training-mode output, all gradients, the advanced running statistics,
    and the eval-mode output computed with them."""
    rec = golden("autoencoder.pt")[tag]
    kind, act_name = tag.split(".")
    y, x, leaves = _ae_run(kind, rec, act_name)
    close(y, rec["y"], 2e-4, 2e-5)
    names = [k for k, v in leaves.items() if v.requires_grad]
    assert names == list(rec["gdigest"])
    grads = torch.autograd.grad(y, [x] + [leaves[k] for k in names], hash_normal(tuple(y.shape), rec["gy_seed"]))
    close(grads[0], rec["gx"], 2e-3, 2e-4)
    for k, g in zip(names, grads[1:]):
        _digest_close(g, rec["gdigest"][k], 5e-3)
        if rec["gparams"][k] is not None:
            close(g, rec["gparams"][k], 5e-3, 5e-4)
    for k, v in rec["buffers_after"].items():
        close(leaves[k].detach(), v, 1e-5, 1e-6) if v.dtype.is_floating_point else None
        assert v.dtype.is_floating_point or int(leaves[k]) == int(v) == 1
    # eval mode with the statistics the training step left behind
    with torch.no_grad():
        p = dict(O.hash_state(rec["shapes"], rec["seed"]), **rec["buffers_after"])
        act = O.act_fn(act_name, p)
        xe = rec["x"]
        ye = {"down": lambda: O.ae_down_block(p, "", xe, act, O.DropMasks(), False), "up": lambda: O.ae_up_block(p, "", xe, act, O.DropMasks(), False),
              "encoder": lambda: O.ae_encoder(p, xe, 2, act, O.DropMasks(), False), "decoder": lambda: O.ae_decoder(p, xe, 2, act, O.DropMasks(), False)}[kind]()
    close(ye, rec["y_eval"], 2e-4, 2e-5)
