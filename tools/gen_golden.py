#!/usr/bin/env python3
"""Generate tests/golden/*.pt by running the REFERENCE implementation on CPU (fp32).

Runs only in the build container, where /root/reference is mounted; the fixtures it writes are
plain tensors (inputs and expected outputs) and are what travels to the GPU box.  The reference
is imported unmodified.  Two third-party modules it imports at module top but never uses on this
path (IPython.display, torchvision.transforms - only the final PIL conversion of DDPM.sample
touches them) are absent from the image, so empty placeholder modules are registered for them.

Weights are NOT stored: they are regenerated from oracle.ref_cpu.hash_params (an integer hash of
the element index), loaded into the reference modules with load_state_dict.  Noise tensors come
from hash_normal in the same way.

    python tools/gen_golden.py            # writes tests/golden/
"""
from __future__ import annotations

import contextlib
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DDPM_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")

for _name in ("IPython", "IPython.display", "torchvision", "torchvision.transforms"):
    sys.modules.setdefault(_name, types.ModuleType(_name))
sys.modules["IPython.display"].HTML = object
import matplotlib  # noqa: E402

matplotlib.use("Agg")
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

import torch  # noqa: E402
import yaml  # noqa: E402

from neural_nets.denoise_models import ddpm as ref_ddpm, unets as ref_unets  # noqa: E402
from neural_nets.denoise_models import constituent_blocks as ref_blocks  # noqa: E402
from neural_nets.conv_block import SeparableInvertResidual  # noqa: E402
from losses.ELBO import ELBOLoss  # noqa: E402

from oracle import ref_cpu as O  # noqa: E402
from oracle.noise import hash_normal, hash_uniform  # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(0)
META = {"torch": str(torch.__version__), "reference": "quyongkeomut/Diffusion-Model @ /root/reference"}


def yaml_cfg():
    with open(os.path.join(REF, "experiments_setup/configs/DDPM.yaml")) as f:
        return yaml.safe_load(f)


def build_ddpm(img_channels=3, activation="gelu", drop_p=0.0, stages=None, latent=None, seed=1, head_gain=1.0):
    y = yaml_cfg()
    kw = dict(y["DENOISE_MODEL_CONFIGS"])
    kw["activation"] = activation
    kw["drop_p"] = drop_p
    if stages is not None:
        kw["stages_channels"] = list(stages)
    latent = latent or y["LATENT_DIM"]
    m = ref_ddpm.DDPM(ref_unets.BaseUNet, **kw, is_conditional=False, img_channels=img_channels,
                      reconstruction_method="mse", latent_dim=latent, device="cpu")
    cfg = O.UNetCfg(img_channels=img_channels, stages_channels=tuple(kw["stages_channels"]), latent_dim=latent,
                    expand_factor=kw["expand_factor"], drop_p=drop_p, t_embed_dim=kw["t_embed_dim"],
                    activation=activation, base_groups_norm=kw["base_groups_norm"])
    params = O.hash_params(cfg, seed=seed, head_gain=head_gain)
    missing = m.load_state_dict(params, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert list(m.state_dict().keys()) == list(params.keys()), "key order differs from reference registration order"
    return m, cfg, params


@contextlib.contextmanager
def patched_noise(seed_base):
    """Make the reference's randn_like / randn draw from hash_normal, numbered in call order."""
    counter = {"i": 0}
    old_like, old_randn = ref_ddpm.randn_like, ref_ddpm.randn

    def like(x):
        counter["i"] += 1
        return hash_normal(tuple(x.shape), seed_base + counter["i"])

    def randn(shape, device=None):
        counter["i"] += 1
        return hash_normal(tuple(shape), seed_base + counter["i"])

    ref_ddpm.randn_like, ref_ddpm.randn = like, randn
    try:
        yield counter
    finally:
        ref_ddpm.randn_like, ref_ddpm.randn = old_like, old_randn


@contextlib.contextmanager
def patched_dropout(seed_base, keep):
    """Make nn.Dropout multiply by hash-derived keep masks, numbered in call order."""
    import torch.nn.functional as F
    counter = {"i": 0}
    old = F.dropout

    def drop(x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        m = (hash_uniform(tuple(x.shape), seed_base + counter["i"]) < keep).float() / keep
        counter["i"] += 1
        return x * m

    F.dropout = drop
    try:
        yield counter
    finally:
        F.dropout = old


def grad_digest(g: torch.Tensor):
    """Compact fingerprint of a gradient tensor: l2, sum, and 16 fixed-position samples."""
    flat = g.detach().reshape(-1).double()
    n = flat.numel()
    idx = (torch.arange(16, dtype=torch.int64) * 2654435761 % max(n, 1))
    return {"l2": flat.norm().item(), "sum": flat.sum().item(), "idx": idx, "val": flat[idx].float()}


def save(name, obj):
    os.makedirs(OUT, exist_ok=True)
    obj = dict(obj)
    obj["meta"] = META
    path = os.path.join(OUT, name)
    torch.save(obj, path)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ------------------------------------------------------------------------------------------------
def gen_schedule_and_ops():
    m, cfg, params = build_ddpm()
    names = ["alphas_bar", "sqrt_alphas_bar", "square_betas", "one_minus_alphas_bar", "square_one_minus_alphas_bar",
             "sqrt_one_minus_alphas_bar", "x_previous_coef", "x0_pred_coef", "bw_std"]
    sched = {k: getattr(m, k).clone() for k in names}
    save("schedule_T120_b025.pt", {"T": 120, "beta": 0.25, "tables": sched})

    n, c, h, w = 6, 3, 8, 8
    x0 = hash_uniform((n, c, h, w), 11) * 2 - 1
    t = torch.tensor([1, 2, 17, 60, 119, 120])
    with patched_noise(100):
        x_t = m.q(x0, t)
    eps = hash_normal((n, c, h, w), 101)
    x0_pred = torch.tanh(hash_normal((n, c, h, w), 12))
    with patched_noise(200):
        x_prev = m.reverse_q(x_t, x0_pred, t)
    z = hash_normal((n, c, h, w), 201)
    crit_mse, crit_bce, crit_plain = ELBOLoss("mse"), ELBOLoss("bce"), ELBOLoss("mse", use_elbo=False)
    xp = x0_pred.clone().requires_grad_(True)
    l_mse = crit_mse(m, t, xp, x0)["Loss"]
    (g_mse,) = torch.autograd.grad(l_mse, xp)
    save("diffusion_ops.pt", {
        "x0": x0, "t": t, "eps": eps, "x_t": x_t, "x0_pred": x0_pred, "z": z, "x_prev": x_prev,
        "loss_mse": l_mse.detach(), "grad_mse": g_mse,
        "loss_bce": crit_bce(m, t, x0_pred, (x0 + 1) / 2)["Loss"].detach(),
        "loss_plain": crit_plain(m, t, x0_pred, x0)["Loss"].detach(),
        "loss_mse_ddpflag": crit_mse(types.SimpleNamespace(module=m), t, x0_pred, x0, is_ddp=True)["Loss"].detach(),
    })


def gen_blocks():
    out = {}
    for act in ("gelu", "silu"):
        kw = dict(expand_factor=2, drop_p=0.0, base_groups_norm=4, activation=act, initializer="he_normal")
        # --- SeparableInvertResidual, C=32 @ 8x8
        torch.manual_seed(1)
        blk = SeparableInvertResidual(32, 32, **kw).eval()
        sd = {k: O.hash_fill(v.shape, 7 + i, 0.25, 1.0 if (k.endswith("2.weight")) else 0.0) for i, (k, v) in enumerate(blk.state_dict().items())}
        blk.load_state_dict(sd)
        x = hash_normal((2, 32, 8, 8), 31).requires_grad_(True)
        y = blk(x)
        gy = hash_normal(tuple(y.shape), 32)
        grads = torch.autograd.grad(y, [x] + list(blk.parameters()), gy)
        out[f"sir.{act}"] = {"params": sd, "x": x.detach(), "y": y.detach(), "gy": gy, "gx": grads[0],
                             "gparams": dict(zip(sd.keys(), grads[1:]))}
        # --- DownBlock 32 -> 64 @ 8x8 -> 4x4
        blk = ref_blocks.DownBlock(in_channels=32, out_channels=64, is_conditional=False, **kw).eval()
        sd = {k: O.hash_fill(v.shape, 57 + i, 0.15, 1.0 if (v.dim() == 1 and k.endswith("weight")) else 0.0) for i, (k, v) in enumerate(blk.state_dict().items())}
        blk.load_state_dict(sd)
        x = hash_normal((2, 32, 8, 8), 41).requires_grad_(True)
        y = blk(x)
        gy = hash_normal(tuple(y.shape), 42)
        grads = torch.autograd.grad(y, [x] + list(blk.parameters()), gy)
        out[f"down.{act}"] = {"params": sd, "x": x.detach(), "y": y.detach(), "gy": gy, "gx": grads[0],
                              "gparams": dict(zip(sd.keys(), grads[1:]))}
        # --- UpBlock 64 -> 32 @ 4x4 -> 8x8
        blk = ref_blocks.UpBlock(in_channels=64, out_channels=32, is_conditional=False, **kw).eval()
        sd = {k: O.hash_fill(v.shape, 157 + i, 0.15, 1.0 if (v.dim() == 1 and k.endswith("weight")) else 0.0) for i, (k, v) in enumerate(blk.state_dict().items())}
        blk.load_state_dict(sd)
        x = hash_normal((2, 64, 4, 4), 51).requires_grad_(True)
        skip = hash_normal((2, 64, 4, 4), 52).requires_grad_(True)
        y = blk(x, skip)
        gy = hash_normal(tuple(y.shape), 53)
        grads = torch.autograd.grad(y, [x, skip] + list(blk.parameters()), gy)
        out[f"up.{act}"] = {"params": sd, "x": x.detach(), "skip": skip.detach(), "y": y.detach(), "gy": gy,
                            "gx": grads[0], "gskip": grads[1], "gparams": dict(zip(sd.keys(), grads[2:]))}
        # --- EmbedBlock 32 -> 64 on sinusoidal(t/100)
        emb = ref_blocks.EmbedBlock(32, 64, act, "he_normal")
        sd = {k: O.hash_fill(v.shape, 257 + i, 0.2) for i, (k, v) in enumerate(emb.state_dict().items())}
        emb.load_state_dict(sd)
        t = torch.tensor([1, 50, 120])
        sin = ref_blocks.SinusoidalPositionEmbedBlock(32)(t.float() / 100)
        out[f"embed.{act}"] = {"params": sd, "t": t, "sin": sin, "y": emb(sin).detach()}
    save("blocks.pt", out)


def gen_sir_variants():
    """The forms of SeparableInvertResidual the reference's UNet never builds (conv_block.py:51-54,84-86): other output width,
    stride 2 in the depthwise convolution; no residual add in either."""
    out = {}
    kw = dict(expand_factor=2, drop_p=0.0, base_groups_norm=4, activation="gelu", initializer="he_normal")
    for tag, cin, cout, stride in (("c32to64.s1", 32, 64, 1), ("c32to64.s2", 32, 64, 2), ("c64.s2", 64, 64, 2)):
        blk = SeparableInvertResidual(cin, cout, stride=stride, **kw).eval()
        assert not blk.is_residual
        sd = {k: O.hash_fill(v.shape, 307 + i, 0.25, 1.0 if (k.endswith("2.weight")) else 0.0) for i, (k, v) in enumerate(blk.state_dict().items())}
        blk.load_state_dict(sd)
        x = hash_normal((3, cin, 8, 8), 331).requires_grad_(True)
        y = blk(x)
        gy = hash_normal(tuple(y.shape), 332)
        grads = torch.autograd.grad(y, [x] + list(blk.parameters()), gy)
        out[tag] = {"params": sd, "x": x.detach(), "y": y.detach(), "gy": gy, "gx": grads[0], "gparams": dict(zip(sd.keys(), grads[1:])),
                    "in_channels": cin, "out_channels": cout, "stride": stride}
    save("sir_variants.pt", out)


def gen_unet_full():
    out = {}
    for tag, c, act in (("c3.gelu", 3, "gelu"), ("c1.gelu", 1, "gelu"), ("c3.silu", 3, "silu")):
        m, cfg, params = build_ddpm(img_channels=c, activation=act)
        m.eval()
        n = 2
        x0 = hash_uniform((n, c, 32, 32), 301) * 2 - 1
        t = torch.tensor([7, 93])
        with patched_noise(400):
            x_t = m.q(x0, t)
        taps = {}
        hooks = [
            m.model.encoder.latent_layers.register_forward_hook(lambda mod, i, o: taps.__setitem__("enc.latent", o.detach())),
            m.model.decoder.out_layers[2].register_forward_hook(lambda mod, i, o: taps.__setitem__("dec.logits", o.detach())),
            m.model.encoder.layers[0].register_forward_hook(lambda mod, i, o: taps.__setitem__("enc.stage0", o.detach())),
            m.model.decoder.layers[3].register_forward_hook(lambda mod, i, o: taps.__setitem__("dec.stage3", o.detach())),
        ]
        pred = m(x_t, t)
        for hk in hooks:
            hk.remove()
        loss = ELBOLoss("mse")(m, t, pred, x0)["Loss"]
        loss.backward()
        digests = {k: grad_digest(v.grad) for k, v in m.named_parameters()}
        out[tag] = {"x0": x0, "t": t, "x_t": x_t, "pred": pred.detach(), "loss": loss.detach(),
                    "taps": taps, "grad_digest": digests, "n_params": sum(p.numel() for p in m.parameters())}
    # train-mode forward with dropout active (masks are hash-derived, site-numbered in call order)
    m, cfg, params = build_ddpm(img_channels=3, activation="gelu", drop_p=0.1)
    m.train()
    x0 = hash_uniform((2, 3, 32, 32), 301) * 2 - 1
    t = torch.tensor([7, 93])
    with patched_noise(400):
        x_t = m.q(x0, t)
    with patched_dropout(9000, keep=0.9) as cnt:
        pred = m(x_t, t)
    loss = ELBOLoss("mse")(m, t, pred, x0)["Loss"]
    loss.backward()
    out["c3.gelu.dropout"] = {"x0": x0, "t": t, "pred": pred.detach(), "loss": loss.detach(), "sites": cnt["i"],
                              "grad_digest": {k: grad_digest(v.grad) for k, v in m.named_parameters()}}
    save("unet_full.pt", out)


def gen_sample_chain():
    m, cfg, params = build_ddpm(img_channels=3, activation="gelu", head_gain=0.3)
    m.eval()
    snaps = {}
    orig = m.reverse_q
    step = {"k": 0}

    def spy(x_t, x0p, times):
        r = orig(x_t, x0p, times)
        step["k"] += 1
        if step["k"] in (1, 2, 30, 60, 119, 120):
            snaps[step["k"]] = r.clone()
        return r

    m.reverse_q = spy
    # DDPM.sample ends by converting to PIL through torchvision; stop before that by making
    # to_image the identity (it is a host-side visualisation helper, not arithmetic).
    old_to_image = ref_ddpm.to_image
    ref_ddpm.to_image = lambda x: x
    try:
        with patched_noise(5000):
            imgs = m.sample(2, shape=(32, 32))
    finally:
        ref_ddpm.to_image = old_to_image
    final = torch.stack(imgs)
    save("sample_chain.pt", {"head_gain": 0.3, "N": 2, "shape": (32, 32), "noise_seed_base": 5000, "snapshots": snaps,
                             "final_unit": final, "note": "call k of randn/randn_like uses hash_normal(shape, 5000+k); k=1 is X_T"})


def gen_conditional():
    torch.manual_seed(3)
    c, depth = 128, 2
    mod = ref_blocks.ConditionalModule(c, "gelu", depth, 0.0, c, c).eval()
    sd = {}
    for i, (k, v) in enumerate(mod.state_dict().items()):
        if v.dim() >= 2:
            sd[k] = O.hash_fill(v.shape, 700 + i, (3.0 / v.shape[-1]) ** 0.5)
        elif "norm" in k and k.endswith("weight"):
            sd[k] = O.hash_fill(v.shape, 700 + i, 0.2, 1.0)
        else:
            sd[k] = O.hash_fill(v.shape, 700 + i, 0.1)
    mod.load_state_dict(sd)
    q = hash_normal((2, 64, c), 801)
    with torch.no_grad():
        y_self = mod(q, q)
    out = {"C": c, "depth": depth, "heads": 8, "param_seed_base": 700,
           "shapes": {k: tuple(v.shape) for k, v in sd.items()}, "q": q, "y_self": y_self, "cases": {}}
    # forward + gradients: self-attention at the 16x16 stage widths, cross-attention with a packed in_proj_weight
    # (kdim == C) and with separate q/k/v projection weights (kdim != C)
    for name, cc, dep, s_q, s_k, kdim, act in (("self_c128_s256", 128, 1, 256, None, None, "gelu"), ("self_c32_s256", 32, 1, 256, None, None, "silu"),
                                               ("cross_packed", 128, 2, 64, 40, 128, "gelu"), ("cross_separate", 64, 1, 64, 24, 48, "gelu")):
        kd = cc if kdim is None else kdim
        mod = ref_blocks.ConditionalModule(cc, act, dep, 0.0, kd, kd).eval()
        params = O.hash_conditional_params("", cc, dep, kd, seed_base=900)
        assert list(mod.state_dict()) == list(params), (list(mod.state_dict()), list(params))
        mod.load_state_dict(params)
        qq = hash_normal((2, s_q, cc), 811).requires_grad_(True)
        cond = qq if s_k is None else hash_normal((2, s_k, kd), 812).requires_grad_(True)
        y = mod(qq, cond)
        gy = hash_normal(tuple(y.shape), 813)
        leaves = [qq] + ([] if s_k is None else [cond]) + list(mod.parameters())
        grads = torch.autograd.grad(y, leaves, gy)
        case = {"C": cc, "depth": dep, "S": s_q, "Sk": s_k, "kdim": kd, "activation": act, "param_seed_base": 900, "q_seed": 811,
                "cond_seed": 812, "gy_seed": 813, "y": y.detach(), "dq": grads[0]}
        rest = grads[1:]
        if s_k is not None:
            case["dcond"], rest = rest[0], rest[1:]
        case["dparams"] = {k: g for (k, _), g in zip(mod.named_parameters(), rest)}
        out["cases"][name] = case
    save("conditional_module.pt", out)


def gen_beta_activations():
    """swish / hardswish (one learnable scalar per activation module, activations.py:25-63): an inverted-residual block, an
    EmbedBlock, a self-attention ConditionalModule and a small whole network, forward and all gradients."""
    out = {}
    for act in ("swish", "hardswish"):
        kw = dict(expand_factor=2, drop_p=0.0, base_groups_norm=4, activation=act, initializer="he_normal")
        blk = SeparableInvertResidual(32, 32, **kw).eval()
        sd = {k: (O.hash_fill((1,), 7 + i, 0.3, 1.0).reshape(()) if k.endswith("beta") else
                  O.hash_fill(v.shape, 7 + i, 0.25, 1.0 if (k.endswith("2.weight")) else 0.0)) for i, (k, v) in enumerate(blk.state_dict().items())}
        blk.load_state_dict(sd)
        x = hash_normal((2, 32, 8, 8), 31).requires_grad_(True)
        y = blk(x)
        gy = hash_normal(tuple(y.shape), 32)
        grads = torch.autograd.grad(y, [x] + list(blk.parameters()), gy)
        out[f"sir.{act}"] = {"params": sd, "x": x.detach(), "y": y.detach(), "gy": gy, "gx": grads[0],
                             "gparams": dict(zip(sd.keys(), grads[1:]))}
        emb = ref_blocks.EmbedBlock(32, 64, act, "he_normal")
        sd = {k: (O.hash_fill((1,), 257 + i, 0.3, 1.0).reshape(()) if k.endswith("beta") else O.hash_fill(v.shape, 257 + i, 0.2))
              for i, (k, v) in enumerate(emb.state_dict().items())}
        emb.load_state_dict(sd)
        t = torch.tensor([1, 50, 120])
        sin = ref_blocks.SinusoidalPositionEmbedBlock(32)(t.float() / 100)
        ye = emb(sin)
        gye = hash_normal(tuple(ye.shape), 33)
        ge = torch.autograd.grad(ye, list(emb.parameters()), gye)
        out[f"embed.{act}"] = {"params": sd, "t": t, "sin": sin, "y": ye.detach(), "gy": gye, "gparams": dict(zip(sd.keys(), ge))}
        # self-attention ConditionalModule, C = 32, 64 tokens
        mod = ref_blocks.ConditionalModule(32, act, 1, 0.0, 32, 32).eval()
        params = O.hash_conditional_params("", 32, 1, 32, seed_base=900, beta=True)
        assert list(mod.state_dict()) == list(params), (list(mod.state_dict()), list(params))
        mod.load_state_dict(params)
        qq = hash_normal((2, 64, 32), 811).requires_grad_(True)
        yc = mod(qq, qq)
        gyc = hash_normal(tuple(yc.shape), 813)
        gc = torch.autograd.grad(yc, [qq] + list(mod.parameters()), gyc)
        out[f"conditional.{act}"] = {"C": 32, "S": 64, "param_seed_base": 900, "q_seed": 811, "gy_seed": 813, "y": yc.detach(), "dq": gc[0],
                                     "dparams": {k: g for (k, _), g in zip(mod.named_parameters(), gc[1:])}}
        # whole network: stages [32, 64], latent 64, 1 channel, 16x16
        m, cfg, params = build_ddpm(img_channels=1, activation=act, stages=(32, 64), latent=64)
        m.eval()
        x0 = hash_uniform((2, 1, 16, 16), 301) * 2 - 1
        t = torch.tensor([7, 93])
        with patched_noise(400):
            x_t = m.q(x0, t)
        pred = m(x_t, t)
        loss = ELBOLoss("mse")(m, t, pred, x0)["Loss"]
        loss.backward()
        out[f"unet.{act}"] = {"x0": x0, "t": t, "x_t": x_t, "pred": pred.detach(), "loss": loss.detach(),
                              "grad_digest": {k: grad_digest(v.grad) for k, v in m.named_parameters()},
                              "beta_grads": {k: v.grad.clone() for k, v in m.named_parameters() if k.endswith(".beta")},
                              "n_params": sum(p.numel() for p in m.parameters())}
    save("beta_activations.pt", out)


def _load_reference_autoencoders():
    """The reference's autoencoder classes.  Its package neural_nets/autoencoders/__init__.py imports a module that does not
    exist, so the two source files are loaded by file path under their own dotted names (the package entry is an empty
    namespace that only carries the directory)."""
    import importlib.util
    base = os.path.join(REF, "neural_nets", "autoencoders")
    pkg = types.ModuleType("neural_nets.autoencoders")
    pkg.__path__ = [base]
    sys.modules["neural_nets.autoencoders"] = pkg
    mods = {}
    for name in ("constituent_blocks", "base_ae"):
        full = f"neural_nets.autoencoders.{name}"
        spec = importlib.util.spec_from_file_location(full, os.path.join(base, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[full] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods["constituent_blocks"], mods["base_ae"]


def _ae_record(mod, seed, x, gy_seed, small=4096):
    """state from oracle.hash_state over the module's own state_dict shapes; one training-mode forward + backward (batch
    statistics, running statistics advance), then an eval-mode forward with the advanced statistics"""
    shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
    mod.load_state_dict(O.hash_state(shapes, seed))
    mod.train()
    xr = x.clone().requires_grad_(True)
    y = mod(xr)
    gy = hash_normal(tuple(y.shape), gy_seed)
    names = [k for k, _ in mod.named_parameters()]
    grads = torch.autograd.grad(y, [xr] + list(mod.parameters()), gy)
    rec = {"shapes": shapes, "seed": seed, "x": x, "y": y.detach(), "gy_seed": gy_seed, "gx": grads[0],
           "gparams": {k: (g if g.numel() <= small else None) for k, g in zip(names, grads[1:])},
           "gdigest": {k: grad_digest(g) for k, g in zip(names, grads[1:])},
           "buffers_after": {k: v.clone() for k, v in mod.named_buffers()}}
    mod.eval()
    with torch.no_grad():
        rec["y_eval"] = mod(x)
    return rec


def gen_autoencoder():
    """The autoencoder side of the latent-diffusion configuration: Down/Up blocks (bias-free convolutions, BatchNorm2d) and the
    Encoder / Decoder stacks, at channel counts the MFMA kernels take (the reference's own dict configs for them are stale)."""
    blocks, base = _load_reference_autoencoders()
    out = {}
    for act in ("gelu", "hardswish"):
        kw = dict(expand_factor=2, drop_p=0.0, activation=act, initializer="he_uniform")
        out[f"down.{act}"] = _ae_record(blocks.DownBlock(32, 64, **kw), 11, hash_normal((2, 32, 8, 8), 61), 62)
        out[f"up.{act}"] = _ae_record(blocks.UpBlock(64, 32, **kw), 12, hash_normal((2, 64, 4, 4), 63), 64)
    kw = dict(expand_factor=2, drop_p=0.0, activation="hardswish", initializer="he_uniform")
    out["encoder.hardswish"] = _ae_record(base.Encoder(3, (64, 128), 128, **kw), 13, hash_uniform((2, 3, 16, 16), 65) * 2 - 1, 66)
    out["decoder.hardswish"] = _ae_record(base.Decoder(3, 128, (128, 64), **kw), 14, hash_normal((2, 128, 2, 2), 67), 68)
    save("autoencoder.pt", out)


if __name__ == "__main__":
    only = sys.argv[1:]
    for fn in (gen_schedule_and_ops, gen_blocks, gen_sir_variants, gen_unet_full, gen_sample_chain, gen_conditional, gen_beta_activations, gen_autoencoder):
        if not only or fn.__name__ in only:
            fn()
