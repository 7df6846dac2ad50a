"""Host-side logic that needs no GPU: the product modules expose exactly the reference's parameters (names, shapes,
registration order), the diffusion schedule matches the reference's tables, and the product refuses to run on the CPU
instead of silently falling back."""
import pytest
import torch

import dm_amd_bootstrap  # noqa: F401
from dm_amd.neural_nets.conv_block import SeparableInvertResidual
from dm_amd.neural_nets.denoise_models import ddpm as dm_ddpm, unets as dm_unets
from dm_amd.neural_nets.denoise_models.constituent_blocks import DownBlock, EmbedBlock, UpBlock
from oracle import ref_cpu as O

YAML = dict(stages_channels=[32, 64, 128, 256], expand_factor=2, drop_p=0.1, T=120, beta=0.25, t_embed_dim=32, activation="gelu",
            base_groups_norm=4, initializer="he_normal", dtype=None)


def build(c=3):
    return dm_ddpm.DDPM(dm_unets.BaseUNet, **YAML, is_conditional=False, img_channels=c, reconstruction_method="mse", latent_dim=512,
                        device="cpu")


def test_state_dict_matches_reference_parameter_list():
    m = build(3)
    ref = O.param_shapes(O.UNetCfg(img_channels=3))
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert list(got.keys()) == list(ref.keys()) and len(got) == 350
    assert all(got[k] == tuple(ref[k]) for k in ref)
    assert sum(v.numel() for v in m.state_dict().values()) == 6857763


def test_schedule_attributes_match_reference_tables(golden):
    m = build(1)
    s = dict(golden("schedule_T120_b025.pt")["tables"])
    for name in ("alphas_bar", "sqrt_one_minus_alphas_bar", "bw_std"):
        torch.testing.assert_close(getattr(m, name).detach().cpu().reshape(-1), s[name].reshape(-1), rtol=2e-7, atol=1e-9, equal_nan=True)
    assert m.T == 120


def test_block_modules_expose_the_reference_keys(golden):
    b = golden("blocks.pt")
    kw = dict(expand_factor=2, drop_p=0.0, base_groups_norm=4, initializer="he_normal", activation="gelu")
    assert list(SeparableInvertResidual(32, 32, **kw).state_dict()) == list(b["sir.gelu"]["params"])
    assert list(DownBlock(in_channels=32, out_channels=64, is_conditional=False, **kw).state_dict()) == list(b["down.gelu"]["params"])
    assert list(UpBlock(in_channels=64, out_channels=32, is_conditional=False, **kw).state_dict()) == list(b["up.gelu"]["params"])
    assert list(EmbedBlock(32, 64, "gelu", "he_normal").state_dict()) == list(b["embed.gelu"]["params"])


def test_no_cpu_fallback():
    m = build(1)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 32, 32), torch.ones(1, dtype=torch.long))
    with pytest.raises(RuntimeError):
        SeparableInvertResidual(32, 32, expand_factor=2, activation="gelu")(torch.zeros(1, 32, 8, 8))


def test_learnable_activations_register_the_reference_keys(golden):
    """swish / hardswish own one 0-dim `beta` per activation module (neural_nets/activations.py:25-63 of the reference): the
    modules here register them under the reference's state_dict keys, in its registration order, initialised to 1.  The
    expected key lists are those of the reference modules the fixture was generated from (tools/gen_golden.py), and
    oracle.ref_cpu.param_shapes, which that generator checked against the reference DDPM's state_dict."""
    from dm_amd.neural_nets.activations import activation_code, has_scalar
    from dm_amd.neural_nets.denoise_models.constituent_blocks import ConditionalModule
    from oracle import ref_cpu as O
    for name in ("gelu", "silu", "relu", "relu6", "leaky_relu", "selu", "hardsigmoid", "sigmoid", "tanh", "swish", "hardswish"):
        assert activation_code(name) > 0
        assert has_scalar(name) == (name in ("swish", "hardswish"))
    g = golden("beta_activations.pt")
    for act in ("swish", "hardswish"):
        kw = dict(expand_factor=2, drop_p=0.0, base_groups_norm=4, initializer="he_normal", activation=act)
        sir = SeparableInvertResidual(32, 32, **kw)
        assert list(sir.state_dict()) == list(g[f"sir.{act}"]["params"])
        assert sir.state_dict()["expansion.3.beta"].shape == () and float(sir.state_dict()["depthwise.3.beta"]) == 1.0
        assert list(EmbedBlock(32, 64, act, "he_normal").state_dict()) == list(g[f"embed.{act}"]["params"])
        assert list(ConditionalModule(32, act, 1, 0.0, 32, 32).state_dict()) == list(g[f"conditional.{act}"]["dparams"])
        cfg = O.UNetCfg(img_channels=1, stages_channels=(32, 64), latent_dim=64, activation=act)
        m = dm_ddpm.DDPM(dm_unets.BaseUNet, **dict(YAML, activation=act, stages_channels=[32, 64]), is_conditional=False, img_channels=1,
                         reconstruction_method="mse", latent_dim=64, device="cpu")
        want = O.param_shapes(cfg)
        assert list(m.state_dict()) == list(want)
        assert all(tuple(v.shape) == want[k] for k, v in m.state_dict().items())
        assert len(g[f"unet.{act}"]["beta_grads"]) == sum(k.endswith(".beta") for k in want) == 36
        down = DownBlock(in_channels=32, out_channels=64, is_conditional=False, **kw)
        assert [k for k in down.state_dict() if k.endswith("beta")] == ["layers.0.2.beta", "layers.1.expansion.3.beta", "layers.1.depthwise.3.beta",
                                                                        "layers.2.expansion.3.beta", "layers.2.depthwise.3.beta", "layers.3.2.beta"]
        up = UpBlock(in_channels=64, out_channels=32, is_conditional=False, **kw)
        assert [k for k in up.state_dict() if k.endswith("beta") and "layers.0." in k or k == "layers.2.2.beta"] == ["layers.0.2.beta", "layers.2.2.beta"]


def test_reference_error_conventions():
    with pytest.raises(ValueError):
        dm_ddpm.DDPM(dm_unets.BaseUNet, **dict(YAML, initializer="nope"), is_conditional=False, img_channels=1, reconstruction_method="mse",
                     latent_dim=512, device="cpu")
    with pytest.raises(KeyError):
        dm_ddpm.DDPM(dm_unets.BaseUNet, **dict(YAML, activation="nope"), is_conditional=False, img_channels=1, reconstruction_method="mse",
                     latent_dim=512, device="cpu")


def test_pil_resampling_tables_reproduce_pil():
    """The host-side coefficient tables of the GPU input pipeline, applied with PIL's two-pass fixed-point arithmetic in
    numpy, equal PIL's own bilinear resize bit for bit."""
    import numpy as np
    from PIL import Image
    from dm_amd.datasets import pil_bilinear_tables
    rng = np.random.default_rng(1)
    for h0, w0, h, w in [(28, 28, 32, 32), (48, 48, 32, 32), (97, 61, 64, 64)]:
        img = rng.integers(0, 256, (h0, w0), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((w, h), Image.BILINEAR))
        hc, hb, _ = pil_bilinear_tables(w0, w)
        vc, vb, _ = pil_bilinear_tables(h0, h)
        t = img.astype(np.int64)
        t = np.stack([np.clip(((1 << 21) + (t[:, hb[x, 0]:hb[x, 0] + hb[x, 1]] * hc[x, :hb[x, 1]].astype(np.int64)).sum(1)) >> 22, 0, 255)
                      for x in range(w)], axis=1)
        t = np.stack([np.clip(((1 << 21) + (t[vb[y, 0]:vb[y, 0] + vb[y, 1]] * vc[y, :vb[y, 1]].astype(np.int64)[:, None]).sum(0)) >> 22, 0, 255)
                      for y in range(h)], axis=0)
        assert np.array_equal(t, ref)


def test_act_backward_split_choice():
    """host logic of the streaming activation backward: the number of pixel ranges per image divides H*W, grows with the
    image, and is 0 (= use the GroupNorm kernel) for channel counts the kernel does not take"""
    import torch
    from dm_amd.runtime import ops
    for h, w, c in [(32, 32, 32), (16, 16, 64), (6, 10, 16), (64, 64, 32), (1, 1, 8), (128, 128, 32)]:
        p = ops.act_backward_splits(torch.empty(2, h, w, c))
        assert p >= 1 and (h * w) % p == 0 and p <= 64
    assert ops.act_backward_splits(torch.empty(2, 64, 64, 32)) > ops.act_backward_splits(torch.empty(2, 8, 8, 32))
    for c in (4, 24, 128, 256):
        assert ops.act_backward_splits(torch.empty(1, 8, 8, c)) == 0


def test_timestep_oracle_is_uniform():
    """the numpy restatement of the in-kernel timestep draw covers 1..T evenly (chi-square well inside the 99.9 % bound)"""
    import numpy as np
    from oracle.philox import sample_timesteps
    T, n = 120, 240000
    t = sample_timesteps(n, T, 123456789, 3)
    counts = np.bincount(t, minlength=T + 1)[1:]
    assert counts.min() > 0
    chi2 = float(((counts - n / T) ** 2 / (n / T)).sum())
    assert chi2 < 180.0, chi2                      # 119 degrees of freedom: mean 119, 99.9 % quantile ~ 173
    assert not np.array_equal(t[:1000], sample_timesteps(1000, T, 123456790, 3))


def test_bench_launcher_starts_one_rank_per_gpu():
    """`python bench.py --gpus N` starts N ranks itself (train_diff.py:114-126 of the reference spawns one process per GPU);
    a --gpus that disagrees with an existing WORLD_SIZE is refused instead of silently running one rank."""
    import json
    import os
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-launch"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    parent = [d for d in lines if d.get("launcher")]
    # the parent only starts processes: it must never have initialised the GPU runtime (a process that has, and then spawns
    # children that open the GPU, is what a shared GPU host forbids)
    assert len(parent) == 1 and parent[0]["gpu_runtime_initialised"] is False
    rows = sorted((d for d in lines if "RANK" in d), key=lambda d: d["RANK"])
    assert [d["RANK"] for d in rows] == ["0", "1"] and [d["LOCAL_RANK"] for d in rows] == ["0", "1"]
    assert all(d["WORLD_SIZE"] == "2" and d["MASTER_ADDR"] == "127.0.0.1" for d in rows)
    assert rows[0]["MASTER_PORT"] == rows[1]["MASTER_PORT"]
    r = subprocess.run([sys.executable, bench, "--gpus", "1", "--dry-launch"], env=dict(env, WORLD_SIZE="2", RANK="0"), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "disagrees with WORLD_SIZE" in r.stderr


def test_sample_bytes_truncate_like_to_pil_image(golden):
    """DDPM.sample ends in torchvision's to_pil_image (utils/other_utils.py:14-37 of the reference), whose float -> byte
    step is pic.mul(255).byte(): a truncation.  The byte image of the reference chain's final tensor must equal that
    arithmetic applied to the fixture's own [0, 1]-clamped result, bit for bit (torchvision is not installed here, so the
    PIL object itself is not compared: parity of that last hop is unpinned)."""
    g = golden("sample_chain.pt")
    final = g["snapshots"][120]
    want = g["final_unit"].mul(255).byte()
    got = dm_ddpm.DDPM.to_uint8(final.clone())
    assert got.dtype == torch.uint8 and torch.equal(got, want)
    x = torch.tensor([[-1.0, -0.999, 0.0, 0.5, 0.9999, 1.0, 3.0]])
    assert dm_ddpm.DDPM.to_uint8(x).tolist() == [[0, 0, 127, 191, 254, 255, 255]]


def test_autoencoder_modules_expose_the_reference_keys(golden):
    """The autoencoder blocks register the reference's parameters AND BatchNorm2d buffers under its state_dict keys, in its
    order, with its shapes (the fixture's shape tables were read off the reference modules, tools/gen_golden.py)."""
    from dm_amd.neural_nets.autoencoders import Decoder, DownBlock as AeDown, Encoder, UpBlock as AeUp
    g = golden("autoencoder.pt")
    for act in ("gelu", "hardswish"):
        kw = dict(expand_factor=2, drop_p=0.0, activation=act, initializer="he_uniform")
        for mod, tag in ((AeDown(32, 64, **kw), f"down.{act}"), (AeUp(64, 32, **kw), f"up.{act}")):
            sd = mod.state_dict()
            assert list(sd) == list(g[tag]["shapes"]) and all(tuple(v.shape) == tuple(g[tag]["shapes"][k]) for k, v in sd.items())
            assert sd["layers.1.num_batches_tracked"].dtype == torch.long and float(sd["layers.1.running_var"][0]) == 1.0
            assert [k for k, _ in mod.named_parameters()] == list(g[tag]["gdigest"])
    kw = dict(expand_factor=2, drop_p=0.0, activation="hardswish", initializer="he_uniform")
    for mod, tag in ((Encoder(3, (64, 128), 128, **kw), "encoder.hardswish"), (Decoder(3, 128, (128, 64), **kw), "decoder.hardswish")):
        sd = mod.state_dict()
        assert list(sd) == list(g[tag]["shapes"]) and all(tuple(v.shape) == tuple(g[tag]["shapes"][k]) for k, v in sd.items())
    with pytest.raises(ValueError):
        Encoder(3, (16, 24), 24, 2)          # the widths of the reference's stale dict config do not fit the MFMA kernels
