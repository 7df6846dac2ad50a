"""GPU parity of the fused diffusion elementwise kernels against the CPU oracle (through the C ABI)."""
import pytest
import torch

from dm_amd.runtime import hip
from oracle import ref_cpu as O
from oracle.noise import hash_normal, hash_uniform

pytestmark = pytest.mark.gpu


def same(a, b):
    """bit-exact comparison with a short failure message"""
    d = (a.double() - b.double()).abs().max().item()
    assert torch.equal(a, b), f"max abs diff {d:.3e}"


def sched_table(golden):
    """fp32 [T+1][8] table built, the way the host code does, from the schedule tensors the reference produced in
    the build container (linspace/cumprod round differently by one ulp on other CPUs, so bit-exact checks must start
    from the stored tables rather than recompute them here)."""
    s = dict(golden("schedule_T120_b025.pt")["tables"])
    T = 120
    tab = torch.zeros(T + 1, 8)
    t = torch.arange(1, T + 1)
    tab[1:, 0] = s["sqrt_alphas_bar"][t].flatten()
    tab[1:, 1] = s["sqrt_one_minus_alphas_bar"][t].flatten()
    tab[1:, 2] = (s["one_minus_alphas_bar"][t - 1] * s["x_previous_coef"][t]).flatten()
    tab[1:, 3] = (s["sqrt_alphas_bar"][t - 1] * s["x0_pred_coef"][t]).flatten()
    tab[1:, 4] = (s["sqrt_one_minus_alphas_bar"][t - 1] * s["bw_std"][t]).flatten()
    tab[1:, 5] = O.elbo_weight(s, t).flatten()
    return s, tab


def test_q_and_reverse_bit_exact_vs_golden(golden, dev):
    g = golden("diffusion_ops.pt")
    _, tab = sched_table(golden)
    tab = tab.to(dev)
    n, c, h, w = g["x0"].shape
    x0, t, eps = g["x0"].to(dev), g["t"].to(dev), g["eps"].to(dev)
    xt = torch.empty_like(x0)
    xt_nhwc = torch.full((n, h, w, 16), 7.0, device=dev, dtype=torch.bfloat16)
    hip.call("ddpm_q_sample", hip.ptr(x0), hip.ptr(t), hip.ptr(eps), hip.ptr(tab), hip.ptr(xt), hip.ptr(xt_nhwc), 16, n, c, h * w,
             None, 0, hip.stream())
    same(xt.cpu(), g["x_t"])
    same(xt_nhwc[..., :c].float().cpu(), g["x_t"].permute(0, 2, 3, 1).bfloat16().float())
    assert (xt_nhwc[..., c:] == 0).all()
    out = torch.empty_like(x0)
    x0p, z = g["x0_pred"].to(dev), g["z"].to(dev)       # keep references: only raw pointers cross the ABI
    hip.call("ddpm_reverse_step", hip.ptr(xt), hip.ptr(x0p), hip.ptr(t), hip.ptr(z), hip.ptr(tab),
             hip.ptr(out), n, c * h * w, None, 0, 0, 0, hip.stream())
    same(out.cpu(), g["x_prev"])


def test_elbo_fwd_bwd(golden, dev):
    g = golden("diffusion_ops.pt")
    _, tab = sched_table(golden)
    tab = tab.to(dev)
    n, c, h, w = g["x0"].shape
    pred, x0, t = g["x0_pred"].to(dev), g["x0"].to(dev), g["t"].to(dev)
    loss = torch.zeros(1, device=dev)
    hip.call("ddpm_elbo_fwd", hip.ptr(pred), hip.ptr(x0), hip.ptr(t), hip.ptr(tab), hip.ptr(loss), n, c * h * w, 0, 1, hip.stream())
    torch.testing.assert_close(loss.cpu()[0], g["loss_mse"], rtol=1e-5, atol=1e-7)
    gp = torch.empty_like(pred)
    hip.call("ddpm_elbo_bwd", hip.ptr(pred), hip.ptr(x0), hip.ptr(t), hip.ptr(tab), None, hip.ptr(gp), n, c * h * w, 0, 1, hip.stream())
    torch.testing.assert_close(gp.cpu(), g["grad_mse"], rtol=1e-5, atol=1e-9)
    y01 = (x0 + 1) / 2
    hip.call("ddpm_elbo_fwd", hip.ptr(pred), hip.ptr(y01), hip.ptr(t), hip.ptr(tab), hip.ptr(loss), n, c * h * w, 1, 1, hip.stream())
    torch.testing.assert_close(loss.cpu()[0], g["loss_bce"], rtol=1e-5, atol=1e-7)
    hip.call("ddpm_elbo_fwd", hip.ptr(pred), hip.ptr(x0), hip.ptr(t), hip.ptr(tab), hip.ptr(loss), n, c * h * w, 0, 0, hip.stream())
    torch.testing.assert_close(loss.cpu()[0], g["loss_plain"], rtol=1e-5, atol=1e-7)


def test_in_kernel_noise_is_standard_normal_and_seeded(golden, dev):
    n, c, hw = 64, 3, 1024
    _, tab = sched_table(golden)
    tab = tab.to(dev)
    x0 = torch.zeros(n, c, hw, device=dev)
    t = torch.full((n,), 120, device=dev, dtype=torch.long)
    seed = torch.tensor([1234], device=dev, dtype=torch.int64)
    a, b = torch.empty_like(x0), torch.empty_like(x0)
    hip.call("ddpm_q_sample", hip.ptr(x0), hip.ptr(t), None, hip.ptr(tab), hip.ptr(a), None, 0, n, c, hw, hip.ptr(seed), 3, hip.stream())
    hip.call("ddpm_q_sample", hip.ptr(x0), hip.ptr(t), None, hip.ptr(tab), hip.ptr(b), None, 0, n, c, hw, hip.ptr(seed), 3, hip.stream())
    assert torch.equal(a, b)
    z = a / tab[120, 1]
    assert abs(z.mean().item()) < 0.01 and abs(z.std().item() - 1.0) < 0.01
    assert abs((z ** 4).mean().item() - 3.0) < 0.1
    hip.call("ddpm_bump_seed", hip.ptr(seed), hip.stream())
    hip.call("ddpm_q_sample", hip.ptr(x0), hip.ptr(t), None, hip.ptr(tab), hip.ptr(b), None, 0, n, c, hw, hip.ptr(seed), 3, hip.stream())
    assert not torch.equal(a, b)
    assert abs(torch.corrcoef(torch.stack([a.flatten(), b.flatten()]))[0, 1].item()) < 0.01


def test_time_embedding(golden, dev):
    e = golden("blocks.pt")["embed.gelu"]
    t = e["t"].to(dev)
    out = torch.empty(t.numel(), 32, device=dev, dtype=torch.bfloat16)
    hip.call("ddpm_time_embedding", hip.ptr(t), 100.0, 32, hip.ptr(out), t.numel(), hip.stream())
    torch.testing.assert_close(out.float().cpu(), e["sin"], rtol=1e-2, atol=4e-3)


def test_layout_bridges(dev):
    x = hash_normal((3, 3, 8, 8), 5).to(dev)
    nhwc = torch.empty(3, 8, 8, 16, device=dev, dtype=torch.bfloat16)
    hip.call("ddpm_nchw_f32_to_nhwc_bf16", hip.ptr(x), hip.ptr(nhwc), 3, 3, 64, 16, hip.stream())
    assert torch.equal(nhwc[..., :3].permute(0, 3, 1, 2).float(), x.bfloat16().float())
    assert (nhwc[..., 3:] == 0).all()
    back = torch.empty_like(x)
    hip.call("ddpm_nhwc_bf16_to_nchw_f32", hip.ptr(nhwc), hip.ptr(back), 3, 3, 64, 16, hip.stream())
    assert torch.equal(back, x.bfloat16().float())
    y = torch.tanh(x)
    dy = hash_normal((3, 3, 8, 8), 6).to(dev)
    dz = torch.empty_like(nhwc)
    hip.call("ddpm_tanh_bwd_nchw_to_nhwc", hip.ptr(dy), hip.ptr(y), hip.ptr(dz), 3, 3, 64, 16, hip.stream())
    ref = (dy * (1 - y * y)).permute(0, 2, 3, 1)
    torch.testing.assert_close(dz[..., :3].float(), ref, rtol=1e-2, atol=1e-3)


def test_timesteps_match_the_philox_oracle_bit_for_bit(dev):
    """Integer work: every drawn timestep equals the numpy restatement (oracle/philox.py, pinned by Random123's known-answer
    vectors); ragged sizes; a bumped seed gives a different draw."""
    from oracle.philox import sample_timesteps
    for n, T, sd, sid in [(256, 120, 1234, 3), (1, 1000, 2 ** 61 + 12345, 0), (1027, 7, 99, 5)]:
        seed = torch.tensor([sd], device=dev, dtype=torch.int64)
        t = torch.full((n + 3,), -7, device=dev, dtype=torch.long)
        hip.call("ddpm_sample_timesteps", hip.ptr(t), n, T, hip.ptr(seed), sid, hip.stream())
        got = t.cpu().numpy()
        assert (got[n:] == -7).all(), "wrote past the end"
        assert (got[:n] == sample_timesteps(n, T, sd, sid)).all()
        assert got[:n].min() >= 1 and got[:n].max() <= T
    seed = torch.tensor([1234], device=dev, dtype=torch.int64)
    a, b = (torch.empty(4096, device=dev, dtype=torch.long) for _ in range(2))
    hip.call("ddpm_sample_timesteps", hip.ptr(a), 4096, 120, hip.ptr(seed), 3, hip.stream())
    hip.call("ddpm_bump_seed", hip.ptr(seed), hip.stream())
    hip.call("ddpm_sample_timesteps", hip.ptr(b), 4096, 120, hip.ptr(seed), 3, hip.stream())
    assert not torch.equal(a, b)
    counts = torch.bincount(a.cpu(), minlength=121)[1:].float()
    assert counts.min() > 10 and counts.max() < 70          # 4096 / 120 = 34 expected per value


@pytest.mark.parametrize("n,c,h", [(5, 1, 8), (3, 3, 4), (2, 32, 4), (256, 1, 32)])
def test_train_head_equals_the_separate_kernels(dev, n, c, h):
    """ddpm_train_head (timestep draw + forward noising into the stem layout + time embedding + seed advance in one launch)
    against the entry points it stands for, bit for bit: ddpm_sample_timesteps (stream 3), ddpm_q_sample (stream 1, NHWC bf16
    output), ddpm_time_embedding and ddpm_bump_seed; with drawn and with injected timesteps / noise."""
    T, dim, cpad = 120, 32, (c + 15) // 16 * 16
    x0 = (hash_uniform((n, c, h, h), 3) * 2 - 1).to(dev)
    tab = torch.rand(T + 1, 8, device=dev)
    for inject in (False, True):
        seed_a = torch.tensor([0x5EED_1234_5678], device=dev, dtype=torch.long)
        seed_b = seed_a.clone()
        t_in = torch.randint(1, T + 1, (n,), device=dev) if inject else None
        z_in = hash_normal((n, c, h, h), 4).to(dev) if inject else None
        # the separate kernels
        t_ref = t_in if inject else torch.empty(n, device=dev, dtype=torch.long)
        if not inject:
            hip.call("ddpm_sample_timesteps", hip.ptr(t_ref), n, T, hip.ptr(seed_a), 3, hip.stream())
        x_ref = torch.full((n, h, h, cpad), 7.0, device=dev, dtype=torch.bfloat16)
        hip.call("ddpm_q_sample", hip.ptr(x0), hip.ptr(t_ref), hip.ptr(z_in), hip.ptr(tab), None, hip.ptr(x_ref), cpad, n, c, h * h,
                 hip.ptr(seed_a), 1, hip.stream())
        hip.call("ddpm_bump_seed", hip.ptr(seed_a), hip.stream())
        s_ref = torch.empty(n, dim, device=dev, dtype=torch.bfloat16)
        hip.call("ddpm_time_embedding", hip.ptr(t_ref), 100.0, dim, hip.ptr(s_ref), n, hip.stream())
        # one launch
        t_out = torch.empty(n, device=dev, dtype=torch.long)
        x_out = torch.full((n, h, h, cpad), 7.0, device=dev, dtype=torch.bfloat16)
        s_out = torch.empty(n, dim, device=dev, dtype=torch.bfloat16)
        done = torch.zeros(1, device=dev, dtype=torch.int32)
        for _ in range(2):                      # twice: the counter returns to zero, the seed advances once per call
            seed_b.copy_(torch.tensor([0x5EED_1234_5678], device=dev))
            hip.call("ddpm_train_head", hip.ptr(x0), hip.ptr(t_in), hip.ptr(z_in), hip.ptr(tab), hip.ptr(t_out), hip.ptr(x_out), cpad,
                     hip.ptr(s_out), 100.0, dim, n, c, h * h, T, hip.ptr(seed_b), hip.ptr(done), hip.stream())
        assert torch.equal(t_out, t_ref) and int(done) == 0
        assert torch.equal(x_out, x_ref)
        assert torch.equal(s_out, s_ref)
        assert torch.equal(seed_a, seed_b)


def test_timestep_range_check_does_not_stall_and_raises(dev):
    """A timestep outside the schedule raises IndexError as the reference's table look-up does (ddpm.py:46-66 there), without
    a host/device rendezvous per call: the offending call computes with the timestep clamped into range, and the error is
    raised by the next checking call (or by check_timesteps(wait=True))."""
    from dm_amd.neural_nets.denoise_models import ddpm as dm_ddpm, unets as dm_unets
    m = dm_ddpm.DDPM(dm_unets.BaseUNet, stages_channels=[32, 64, 128, 256], expand_factor=2, drop_p=0.0, T=120, beta=0.25,
                     t_embed_dim=32, activation="gelu", base_groups_norm=4, initializer="he_normal", dtype=None,
                     is_conditional=False, img_channels=1, reconstruction_method="mse", latent_dim=512, device=dev).to(dev)
    x = torch.rand(3, 1, 32, 32, device=dev)
    noise = torch.randn(3, 1, 32, 32, device=dev)
    ok = torch.tensor([0, 60, 120], device=dev)
    a = m.q(x, ok, noise=noise)
    m.check_timesteps(wait=True)                                   # nothing pending
    assert torch.equal(a[0], x[0])                                 # t = 0 is the identity (fw_mean_coef[0] = 1, fw_std[0] = 0)
    bad = torch.tensor([5, 121, 7], device=dev)
    b = m.q(x, bad, noise=noise)                                   # computes with 121 clamped to 120
    torch.cuda.synchronize()
    with pytest.raises(IndexError, match="121"):
        m.q(x, ok, noise=noise)
    m.check_timesteps(wait=True)                                   # the flag was cleared by the raise
    c = m.q(x, torch.tensor([5, 120, 7], device=dev), noise=noise)
    assert torch.equal(b, c)
    m.reverse_q(x, x, torch.tensor([0, 3, 4], device=dev), noise=noise)     # the reverse step starts at t = 1
    with pytest.raises(IndexError):
        m.check_timesteps(wait=True)
    # strict mode: the offending call itself raises, as in the reference (one rendezvous per call)
    m.strict_timesteps = True
    with pytest.raises(IndexError, match="121"):
        m.q(x, bad, noise=noise)
    m.q(x, ok, noise=noise)                                        # nothing left pending
    with pytest.raises(IndexError):
        m.reverse_q(x, x, torch.tensor([0, 3, 4], device=dev), noise=noise)


@pytest.mark.parametrize("c", [1, 3])
def test_fused_loss_and_head_gradient_matches_the_separate_kernels(dev, c):
    """ddpm_elbo_tanh_head = ddpm_elbo_fwd + ddpm_elbo_bwd + ddpm_tanh_bwd_nchw_to_nhwc + ddpm_colsum in one launch: the
    gradient of the logits bit for bit, the two sums (loss, bias gradient) to fp32 rounding of their order."""
    from dm_amd.runtime import hip, ops
    n, hw = 37, 32 * 32
    y = torch.tanh(hash_normal((n, c, 32, 32), 71)).to(dev)
    x0 = (hash_uniform((n, c, 32, 32), 72) * 2 - 1).to(dev)
    t = torch.randint(1, 121, (n,), device=dev)
    table = (torch.rand(121, 8, device=dev) + 0.5).contiguous()
    for weighted in (1, 0):
        loss_a, loss_b = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        hip.call("ddpm_elbo_fwd", hip.ptr(y), hip.ptr(x0), hip.ptr(t), hip.ptr(table), hip.ptr(loss_a), n, c * hw, 0, weighted, hip.stream())
        dpred = torch.empty_like(y)
        hip.call("ddpm_elbo_bwd", hip.ptr(y), hip.ptr(x0), hip.ptr(t), hip.ptr(table), None, hip.ptr(dpred), n, c * hw, 0, weighted, hip.stream())
        dz_a = torch.empty(n, 32, 32, 16, device=dev, dtype=torch.bfloat16)
        hip.call("ddpm_tanh_bwd_nchw_to_nhwc", hip.ptr(dpred), hip.ptr(y), hip.ptr(dz_a), n, c, hw, 16, hip.stream())
        bias_a = torch.full((c,), 0.25, device=dev)
        ops.colsum(dz_a, bias_a, c)
        dz_b = torch.full((n, 32, 32, 16), float("nan"), device=dev, dtype=torch.bfloat16)
        bias_b = torch.full((c,), 0.25, device=dev)
        hip.call("ddpm_elbo_tanh_head", hip.ptr(y), hip.ptr(x0), hip.ptr(t), hip.ptr(table), hip.ptr(loss_b), hip.ptr(dz_b), hip.ptr(bias_b),
                 n, c, hw, weighted, hip.stream())
        torch.cuda.synchronize()
        assert torch.equal(dz_a.view(torch.int16), dz_b.view(torch.int16))
        assert abs(loss_a.item() - loss_b.item()) <= 1e-5 * abs(loss_a.item())
        torch.testing.assert_close(bias_b, bias_a, rtol=1e-4, atol=1e-7)
