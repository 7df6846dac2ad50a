"""CPU oracle for the DDPM hot path (TEST INFRASTRUCTURE, not product code).

A functional, fp32, pure-torch restatement of the reference's UNet denoiser,
diffusion schedule, forward noising, ancestral reverse step and ELBO objective.
It exists only so that tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg can check (and time) the HIP path against the reference's
arithmetic.  Nothing in the product package may import this module.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
against tests/golden/*.pt, which tools/gen_golden.py produced by importing the
reference itself (CPU, fp32) in the build container.

Parameters are passed as a flat dict {state_dict key: tensor} using exactly the
reference's key names ("model.encoder.stem_layers.0.weight", ...), so a
reference checkpoint can be evaluated directly.  Each function cites the
reference lines it restates.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

Params = Dict[str, Tensor]

# When True, every tensor the HIP path stores in bf16 (convolution weights, each convolution's output, each
# GroupNorm/activation output, the time-embedding-shifted stage input) is rounded to bf16 here as well, with
# all arithmetic still in fp32.  This "bf16-storage" oracle isolates kernel errors from the precision the
# design states; parity against the plain fp32 oracle is reported separately.
EMULATE_BF16 = False


def _r(x: Tensor) -> Tensor:
    return x.bfloat16().float() if EMULATE_BF16 else x


# --------------------------------------------------------------------------- config
@dataclass
class UNetCfg:
    """Constructor keywords of the denoiser (reference experiments_setup/configs/DDPM.yaml:4-20).

    time_norm is the constant the raw timestep is divided by before the sinusoidal
    embedding.  The reference never forwards the YAML's T to the UNet, so its
    UNetHolder default applies (neural_nets/denoise_models/dm_utils.py:46): 100.
    """
    img_channels: int = 3
    stages_channels: Sequence[int] = (32, 64, 128, 256)
    latent_dim: int = 512
    expand_factor: int = 2
    drop_p: float = 0.1
    t_embed_dim: int = 32
    activation: str = "gelu"
    base_groups_norm: int = 4
    time_norm: float = 100.0
    # Stages whose Down/Up blocks carry a self-attention ConditionalModule at layers[1], the place where the
    # reference's commented-out construction would insert it (constituent_blocks.py:92-101,194-203).  The reference
    # itself never builds it, so anything with a non-empty tuple here is pinned at the ConditionalModule level only.
    attention_stages: Sequence[int] = ()
    attention_depth: int = 1

    @property
    def groups(self) -> int:          # constituent_blocks.py:68-72, conv_block.py:67-71
        return 4 * self.base_groups_norm

    @property
    def latent_groups(self) -> int:   # unets.py:80-84
        return 8 * self.base_groups_norm


BETA_ACTS = ("swish", "hardswish")


def act_fn(name: str, p: Optional[Params] = None):
    """Activations of the reference registry (neural_nets/activations.py:102-114) as act(x, site): `site` is the state_dict
    prefix of the activation module at that place of the network.  The parameter-free ones ignore it; swish and hardswish
    (activations.py:25-63: x * sigmoid(beta x), x * hardsigmoid(beta x)) read their learnable scalar p[site + ".beta"]."""
    if name in BETA_ACTS:
        gate = torch.sigmoid if name == "swish" else F.hardsigmoid
        assert p is not None, f"{name} needs the parameter dictionary"
        return lambda x, site: x * gate(p[site + ".beta"] * x)
    f = _PLAIN_ACTS()[name]
    return lambda x, site=None: f(x)


def _PLAIN_ACTS():
    table = {
        "gelu": lambda x: F.gelu(x, approximate="none"),
        "silu": F.silu,
        "relu": F.relu,
        "relu6": F.relu6,
        "sigmoid": torch.sigmoid,
        "tanh": torch.tanh,
        "hardsigmoid": F.hardsigmoid,
        "leaky_relu": lambda x: F.leaky_relu(x, 0.05),
        "selu": F.selu,
    }
    return table


# --------------------------------------------------------------------------- schedule
def make_schedule(T: int, beta: float = 0.25, device="cpu") -> Dict[str, Tensor]:
    """The nine (T+1,1,1,1) fp32 tables of DDPM.__init__ (ddpm.py:44-62), same op order."""
    betas = torch.linspace(0.0, beta, steps=T + 1, device=device).view(T + 1, 1, 1, 1)
    alphas = 1.0 - betas
    alphas_bar = torch.cumprod(alphas, dim=0)
    one_minus = 1.0 - alphas_bar
    return {
        "betas": betas,
        "alphas": alphas,
        "alphas_bar": alphas_bar,
        "sqrt_alphas_bar": torch.sqrt(alphas_bar),
        "square_betas": betas ** 2,
        "one_minus_alphas_bar": one_minus,
        "square_one_minus_alphas_bar": one_minus ** 2,
        "sqrt_one_minus_alphas_bar": torch.sqrt(one_minus),
        "x_previous_coef": torch.sqrt(alphas) / one_minus,
        "x0_pred_coef": betas / one_minus,
        "bw_std": torch.sqrt(betas / one_minus),
    }


def q_sample(s: Dict[str, Tensor], x0: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """x_t = sqrt(abar_t) x0 + sqrt(1-abar_t) eps   (ddpm.py:102-106)."""
    t = t.int()
    mean = s["sqrt_alphas_bar"][t] * x0
    return mean + s["sqrt_one_minus_alphas_bar"][t] * noise


def reverse_step(s: Dict[str, Tensor], x_t: Tensor, x0_pred: Tensor, t: Tensor, z: Tensor) -> Tensor:
    """x0-form posterior sample (ddpm.py:138-144), same association order."""
    t = t.int()
    mean_prev = (
        s["one_minus_alphas_bar"][t - 1] * s["x_previous_coef"][t] * x_t
        + s["sqrt_alphas_bar"][t - 1] * s["x0_pred_coef"][t] * x0_pred
    )
    return mean_prev + s["sqrt_one_minus_alphas_bar"][t - 1] * s["bw_std"][t] * z


def elbo_weight(s: Dict[str, Tensor], t: Tensor, eps: float = 1e-4) -> Tensor:
    """Per-sample weight w_t of ELBOLoss (losses/ELBO.py:66-67), shape (N,1,1,1)."""
    vars_t = (s["sqrt_one_minus_alphas_bar"][t - 1] * s["bw_std"][t]) ** 2
    return 0.5 * s["square_betas"][t] * s["alphas_bar"][t - 1] / (vars_t * s["square_one_minus_alphas_bar"][t] + eps)


def elbo_loss(s: Dict[str, Tensor], t: Tensor, x_pred: Tensor, x_in: Tensor,
              method: str = "mse", use_elbo: bool = True, eps: float = 1e-4) -> Tensor:
    """ELBOLoss.forward (losses/ELBO.py:61-76) -> 0-d tensor."""
    fn = F.mse_loss if method == "mse" else F.binary_cross_entropy_with_logits
    if not use_elbo:
        return fn(x_pred, x_in)
    w = elbo_weight(s, t, eps)
    return (fn(x_pred, x_in, reduction="none") * w.expand_as(x_in)).mean()


# --------------------------------------------------------------------------- blocks
def sinusoidal(t_norm: Tensor, dim: int) -> Tensor:
    """SinusoidalPositionEmbedBlock.forward (constituent_blocks.py:261-279)."""
    half = dim // 2
    pre = math.log(10000) / (half - 1)
    freq = torch.exp(torch.arange(half, device=t_norm.device) * -pre)
    arg = t_norm.view(-1, 1) * freq[None, :]
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def embed_block(p: Params, pre: str, temb: Tensor, act) -> Tensor:
    """EmbedBlock: Linear -> act -> Linear -> (N,C,1,1)   (constituent_blocks.py:308-328)."""
    h = _r(F.linear(_r(temb), _r(p[pre + "layers.0.weight"]), p[pre + "layers.0.bias"]))
    h = F.linear(_r(act(h, pre + "layers.1")), _r(p[pre + "layers.2.weight"]), p[pre + "layers.2.bias"])
    return h[:, :, None, None]


class DropMasks:
    """Pre-drawn keep masks so a dropout site can be replayed identically by another
    implementation.  masks[i] multiplies the input of the i-th dropout site in call
    order and already contains the 1/(1-p) scale.  Empty => dropout disabled."""

    def __init__(self, masks: Optional[List[Tensor]] = None):
        self.masks = masks or []
        self.i = 0

    def __call__(self, x: Tensor) -> Tensor:
        if not self.masks:
            return x
        m = self.masks[self.i]
        self.i += 1
        return x * m


def conv_gn_act(p: Params, conv: str, gn: str, x: Tensor, groups: int, act, *, stride=1, transposed=False) -> Tensor:
    """Conv(3x3 | convT 2x2 s2) -> GroupNorm -> activation triple (constituent_blocks.py:65-74,166-188)."""
    if transposed:
        y = F.conv_transpose2d(x, _r(p[conv + "weight"]), p[conv + "bias"], stride=2)
    else:
        y = F.conv2d(x, _r(p[conv + "weight"]), p[conv + "bias"], stride=stride, padding=1)
    y = F.group_norm(_r(y), groups, p[gn + "weight"], p[gn + "bias"], eps=1e-5)
    return _r(act(y, gn[:-2] + "2"))          # Sequential(conv, GroupNorm, activation): the activation is module 2


def sir_block(p: Params, pre: str, x: Tensor, groups: int, act, drop: DropMasks, stride: int = 1) -> Tensor:
    """SeparableInvertResidual.forward (conv_block.py:57-136): three conv+GN stages, each fed a
    dropped-out input; the residual adds the un-dropped block input when the block keeps the shape
    (in_channels == out_channels and stride 1, conv_block.py:51-54); `stride` is the depthwise
    convolution's (conv_block.py:84)."""
    e = _r(F.conv2d(_r(drop(x)), _r(p[pre + "expansion.1.weight"]), p[pre + "expansion.1.bias"]))
    e = _r(drop(act(F.group_norm(e, groups, p[pre + "expansion.2.weight"], p[pre + "expansion.2.bias"], eps=1e-5), pre + "expansion.3")))
    c = e.shape[1]
    d = _r(F.conv2d(e, p[pre + "depthwise.1.weight"], p[pre + "depthwise.1.bias"], stride=stride, padding=1, groups=c))
    d = _r(drop(act(F.group_norm(d, groups, p[pre + "depthwise.2.weight"], p[pre + "depthwise.2.bias"], eps=1e-5), pre + "depthwise.3")))
    o = _r(F.conv2d(d, _r(p[pre + "projection.1.weight"]), p[pre + "projection.1.bias"]))
    o = F.group_norm(o, groups, p[pre + "projection.2.weight"], p[pre + "projection.2.bias"], eps=1e-5)
    if o.shape == x.shape and stride == 1:
        o = o + x
    return _r(o)


def _attend(p: Params, pre: str, z: Tensor, depth: int, act) -> Tensor:
    """The isinstance(layer, ConditionalModule) branch of the blocks' forward (constituent_blocks.py:137-143,233-239):
    flatten to tokens, run the module with condition = the tokens themselves, unflatten."""
    n, c, h, w = z.shape
    tok = _r(z).flatten(2).transpose(1, 2)
    tok = conditional_module(p, pre, tok, tok, depth, act)
    return _r(tok.transpose(1, 2).reshape(n, c, h, w))


def down_block(p: Params, pre: str, x: Tensor, groups: int, act, drop: DropMasks, attn_depth: int = 0) -> Tensor:
    """DownBlock.forward (constituent_blocks.py:63-90,135-147); attn_depth > 0 = a ConditionalModule at layers[1]."""
    z = conv_gn_act(p, pre + "layers.0.0.", pre + "layers.0.1.", x, groups, act, stride=2)
    k = 1
    if attn_depth:
        z = _attend(p, pre + "layers.1.", z, attn_depth, act)
        k = 2
    z = sir_block(p, f"{pre}layers.{k}.", z, groups, act, drop)
    z = sir_block(p, f"{pre}layers.{k + 1}.", z, groups, act, drop)
    return conv_gn_act(p, f"{pre}layers.{k + 2}.0.", f"{pre}layers.{k + 2}.1.", z, groups, act)


def up_block(p: Params, pre: str, x: Tensor, skip: Tensor, groups: int, act, drop: DropMasks, attn_depth: int = 0) -> Tensor:
    """UpBlock.forward (constituent_blocks.py:165-193,227-243); attn_depth > 0 = a ConditionalModule at layers[1]."""
    z = torch.cat([x, skip], dim=1)
    z = conv_gn_act(p, pre + "layers.0.0.", pre + "layers.0.1.", z, groups, act)
    k = 1
    if attn_depth:
        z = _attend(p, pre + "layers.1.", z, attn_depth, act)
        k = 2
    z = sir_block(p, f"{pre}layers.{k}.", z, groups, act, drop)
    z = conv_gn_act(p, f"{pre}layers.{k + 1}.0.", f"{pre}layers.{k + 1}.1.", z, groups, act, transposed=True)
    z = sir_block(p, f"{pre}layers.{k + 2}.", z, groups, act, drop)
    return sir_block(p, f"{pre}layers.{k + 3}.", z, groups, act, drop)


# --------------------------------------------------------------------------- UNet
def unet_forward(p: Params, cfg: UNetCfg, x: Tensor, time: Tensor, *, prefix: str = "model.",
                 drop: Optional[DropMasks] = None, taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """BaseUNet.forward = decoder(encoder(x, t), skips, t)   (unets.py:100-132,206-240,250-273).

    taps, when given, is filled with named intermediate tensors (stage outputs, latent,
    pre-Tanh logits) so per-layer parity can be checked."""
    act = act_fn(cfg.activation, p)
    drop = drop or DropMasks()
    g = cfg.groups
    nst = len(cfg.stages_channels)
    enc, dec = prefix + "encoder.", prefix + "decoder."

    def tap(name, v):
        if taps is not None:
            taps[name] = v
        return v

    # ---- encoder (unets.py:121-132)
    temb = sinusoidal(time.float() / cfg.time_norm, cfg.t_embed_dim)
    z = _r(act(_r(F.conv2d(_r(x), _r(p[enc + "stem_layers.0.weight"]), p[enc + "stem_layers.0.bias"], padding=1)), enc + "stem_layers.1"))
    tap("enc.stem", z)
    skips = []
    for i in range(nst):
        z = down_block(p, f"{enc}layers.{i}.", _r(z + embed_block(p, f"{enc}t_embeddings.{i}.", temb, act)), g, act, drop,
                       cfg.attention_depth if i in cfg.attention_stages else 0)
        skips.append(tap(f"enc.stage{i}", z))
    z = _r(F.conv2d(z, _r(p[enc + "latent_layers.0.weight"]), p[enc + "latent_layers.0.bias"], padding=1))
    z = _r(act(F.group_norm(z, cfg.latent_groups, p[enc + "latent_layers.1.weight"], p[enc + "latent_layers.1.bias"], eps=1e-5),
               enc + "latent_layers.2"))
    tap("enc.latent", z)

    # ---- decoder (unets.py:230-240)
    z = conv_gn_act(p, dec + "decoder_input_layers.0.", dec + "decoder_input_layers.1.", z, g, act)
    tap("dec.input", z)
    skips = skips[::-1]
    for i in range(nst):
        z = up_block(p, f"{dec}layers.{i}.", _r(z + embed_block(p, f"{dec}t_embeddings.{i}.", temb, act)), skips[i], g, act, drop,
                     cfg.attention_depth if nst - 1 - i in cfg.attention_stages else 0)
        tap(f"dec.stage{i}", z)
    z = _r(act(_r(F.conv2d(z, _r(p[dec + "out_layers.0.weight"]), p[dec + "out_layers.0.bias"], padding=1)), dec + "out_layers.1"))
    z = F.conv2d(z, _r(p[dec + "out_layers.2.weight"]), p[dec + "out_layers.2.bias"], padding=1)
    tap("dec.logits", z)
    return torch.tanh(z)


def count_dropout_sites(cfg: UNetCfg) -> int:
    """3 per inverted-residual block; 2 per down stage + 3 per up stage (conv_block.py:57-58,77-78,100-101)."""
    return 3 * 5 * len(cfg.stages_channels)


def draw_drop_masks(p: Params, cfg: UNetCfg, x: Tensor, time: Tensor, gen: torch.Generator) -> List[Tensor]:
    """Draw one Bernoulli keep mask (scaled by 1/(1-p)) per dropout site, in call order,
    with the shapes a forward pass presents to each site."""
    shapes: List[torch.Size] = []

    class _Rec(DropMasks):
        def __call__(self, v):
            shapes.append(v.shape)
            return v

    with torch.no_grad():
        unet_forward(p, cfg, x, time, drop=_Rec())
    keep = 1.0 - cfg.drop_p
    return [(torch.rand(s, generator=gen) < keep).float() / keep for s in shapes]


# --------------------------------------------------------------------------- conditional module (never instantiated by the reference UNet)
def conditional_module(p: Params, pre: str, query: Tensor, cond: Tensor, depth: int, act, num_heads: int = 8) -> Tensor:
    """ConditionalModule.forward (constituent_blocks.py:402-437) in eval mode, written with explicit
    attention arithmetic (packed in_proj when kdim == vdim == C, which is the self-attention use)."""
    z = query                                           # (N, S, C)
    n, s, c = z.shape
    dh = c // num_heads
    for i in range(depth):
        m = f"{pre}mhas.{i}."
        b_in = p[m + "in_proj_bias"]
        if m + "in_proj_weight" in p:                   # kdim == vdim == C: one packed weight
            w_in = p[m + "in_proj_weight"]
            wq, wk, wv = w_in[:c], w_in[c:2 * c], w_in[2 * c:]
        else:                                           # separate projections (torch keeps q/k/v_proj_weight then)
            wq, wk, wv = p[m + "q_proj_weight"], p[m + "k_proj_weight"], p[m + "v_proj_weight"]
        q = _r(F.linear(_r(z), _r(wq), b_in[:c]))
        k = _r(F.linear(_r(cond), _r(wk), b_in[c:2 * c]))
        v = _r(F.linear(_r(cond), _r(wv), b_in[2 * c:]))
        sk = cond.shape[1]
        q = q.view(n, s, num_heads, dh).transpose(1, 2)
        k = k.view(n, sk, num_heads, dh).transpose(1, 2)
        v = v.view(n, sk, num_heads, dh).transpose(1, 2)
        att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), dim=-1)
        o = _r((att @ v).transpose(1, 2).reshape(n, s, c))
        o = F.linear(o, _r(p[m + "out_proj.weight"]), p[m + "out_proj.bias"])
        # GroupNorm(1, C) over the (C, S, 1) view == joint norm over C*S per sample (:425-428)
        r = _r(o + z).transpose(1, 2).unsqueeze(-1)
        r = _r(F.group_norm(r, 1, p[f"{pre}mha_norms.{i}.weight"], p[f"{pre}mha_norms.{i}.bias"], eps=1e-5))
        h = _r(F.linear(r.squeeze(-1).transpose(1, 2), _r(p[f"{pre}linears.{i}.weight"]), p[f"{pre}linears.{i}.bias"]))
        h = act(h, f"{pre}activations.{i}").transpose(1, 2).unsqueeze(-1) + r
        h = _r(F.group_norm(h, 1, p[f"{pre}linear_norms.{i}.weight"], p[f"{pre}linear_norms.{i}.bias"], eps=1e-5))
        z = h.squeeze(-1).transpose(1, 2)
    return z


def conditional_shapes(pre: str, c: int, depth: int, kdim: Optional[int] = None, beta: bool = False) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys and shapes of ConditionalModule(c, act, depth, p, kdim, kdim) in registration order
    (constituent_blocks.py:357-399; nn.MultiheadAttention registers in_proj_weight - or q/k/v_proj_weight when
    kdim != embed_dim - then in_proj_bias, then out_proj)."""
    kdim = c if kdim is None else kdim
    out: Dict[str, Tuple[int, ...]] = {}
    for i in range(depth):
        m = f"{pre}mhas.{i}."
        if kdim == c:
            out[m + "in_proj_weight"] = (3 * c, c)
        else:
            out[m + "q_proj_weight"] = (c, c); out[m + "k_proj_weight"] = (c, kdim); out[m + "v_proj_weight"] = (c, kdim)
        out[m + "in_proj_bias"] = (3 * c,); out[m + "out_proj.weight"] = (c, c); out[m + "out_proj.bias"] = (c,)
    for name, shp in (("mha_norms", None), ("linears", (c, c)), ("linear_norms", None)):
        for i in range(depth):
            if shp:
                out[f"{pre}{name}.{i}.weight"] = shp
            else:
                out[f"{pre}{name}.{i}.weight"] = (c,)
            out[f"{pre}{name}.{i}.bias"] = (c,)
    if beta:                                             # self.activations, registered last (constituent_blocks.py:398-399)
        for i in range(depth):
            out[f"{pre}activations.{i}.beta"] = ()
    return out


def hash_conditional_params(pre: str, c: int, depth: int, kdim: Optional[int] = None, seed_base: int = 700, beta: bool = False) -> Params:
    """Deterministic ConditionalModule parameters (the rule tools/gen_golden.py used for conditional_module.pt)."""
    p: Params = {}
    for i, (k, shp) in enumerate(conditional_shapes(pre, c, depth, kdim, beta).items()):
        if k.endswith(".beta"):
            p[k] = hash_fill((1,), seed_base + i, 0.3, 1.0).reshape(())
        elif len(shp) >= 2:
            p[k] = hash_fill(shp, seed_base + i, (3.0 / shp[-1]) ** 0.5)
        elif "norm" in k and k.endswith("weight"):
            p[k] = hash_fill(shp, seed_base + i, 0.2, 1.0)
        else:
            p[k] = hash_fill(shp, seed_base + i, 0.1)
    return p


# --------------------------------------------------------------------------- loops
def train_step_grads(p: Params, cfg: UNetCfg, s: Dict[str, Tensor], x0: Tensor, t: Tensor, noise: Tensor,
                     masks: Optional[List[Tensor]] = None, method: str = "mse") -> Tuple[Tensor, Params]:
    """One micro-step of BaseTrainer._forward_pass + backward (base_trainer.py:196-225):
    q -> UNet -> ELBO -> autograd.  Returns (loss, {name: grad})."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    x_t = q_sample(s, x0, t, noise)
    pred = unet_forward(leaves, cfg, x_t, t, drop=DropMasks(masks))
    loss = elbo_loss(s, t, pred, x0, method)
    names = list(leaves)
    grads = torch.autograd.grad(loss, [leaves[k] for k in names])
    return loss.detach(), dict(zip(names, grads))


@torch.no_grad()
def sample_chain(p: Params, cfg: UNetCfg, s: Dict[str, Tensor], x_T: Tensor, T: int,
                 noises: Sequence[Tensor], method: str = "mse", steps: Optional[int] = None) -> Tensor:
    """DDPM.sample's loop (ddpm.py:153-164) on a given start tensor and given per-step
    noise; returns the tensor before the (x+1)/2 rescale.  steps limits the number of
    iterations (default: all T)."""
    x = x_T
    n = x.shape[0]
    times = torch.full((n,), T, dtype=torch.long)
    k = 0
    while T >= 1 and (steps is None or k < steps):
        x0p = unet_forward(p, cfg, x, times)
        if method == "bce":
            x0p = torch.sigmoid(x0p)
        x = reverse_step(s, x, x0p, times, noises[k])
        T -= 1
        times = times - 1
        k += 1
    return x


def to_unit_range(x: Tensor) -> Tensor:
    """(x+1)/2 then clamp to [0,1]: DDPM.sample's rescale + to_image's min/max (ddpm.py:164; utils/other_utils.py:31-34)."""
    return ((x + 1) / 2).clamp(0.0, 1.0)


# --------------------------------------------------------------------------- deterministic parameters
def hash_fill(shape: Sequence[int], seed: int, scale: float, offset: float = 0.0) -> Tensor:
    """Platform-independent pseudo-random fp32 tensor on [offset-scale, offset+scale): an integer hash
    of the element index (oracle/noise.py).  Gives goldens reproducible weights without storing them
    and without relying on any RNG stream."""
    from oracle.noise import _hash_u32, _numel
    u = _hash_u32(_numel(shape), seed).double() / 4294967296.0 - 0.5
    return (u * (2.0 * scale) + offset).float().view(*shape)


def param_shapes(cfg: UNetCfg, prefix: str = "model.") -> Dict[str, Tuple[int, ...]]:
    """Every state_dict key of DDPM(BaseUNet) with its shape, in registration order
    (unets.py:40-90,146-193; constituent_blocks.py:63-90,165-193,308-317; conv_block.py:57-116)."""
    out: Dict[str, Tuple[int, ...]] = {}
    st = list(cfg.stages_channels)
    te = cfg.t_embed_dim
    has_beta = cfg.activation in BETA_ACTS

    def conv(name, co, ci, k):
        out[name + "weight"] = (co, ci, k, k)
        out[name + "bias"] = (co,)

    def norm(name, c):
        out[name + "weight"] = (c,)
        out[name + "bias"] = (c,)

    def actp(site):
        """the activation module at `site`: swish / hardswish register one 0-dim parameter (activations.py:40,60)"""
        if has_beta:
            out[site + ".beta"] = ()

    def sir(name, c):
        e = int(cfg.expand_factor * c)
        conv(name + "expansion.1.", e, c, 1); norm(name + "expansion.2.", e); actp(name + "expansion.3")
        conv(name + "depthwise.1.", e, 1, 3); norm(name + "depthwise.2.", e); actp(name + "depthwise.3")
        conv(name + "projection.1.", c, e, 1); norm(name + "projection.2.", c)

    def embed(name, c):
        out[name + "layers.0.weight"] = (2 * c, te); out[name + "layers.0.bias"] = (2 * c,)
        actp(name + "layers.1")
        out[name + "layers.2.weight"] = (c, 2 * c); out[name + "layers.2.bias"] = (c,)

    e = prefix + "encoder."
    conv(e + "stem_layers.0.", st[0], cfg.img_channels, 3); actp(e + "stem_layers.1")
    ins = [st[0]] + st[:-1]
    for i, (ci, co) in enumerate(zip(ins, st)):
        b = f"{e}layers.{i}.layers."
        conv(b + "0.0.", co, ci, 3); norm(b + "0.1.", co); actp(b + "0.2")
        k = 1
        if i in cfg.attention_stages:
            out.update(conditional_shapes(b + "1.", co, cfg.attention_depth, beta=has_beta))
            k = 2
        sir(f"{b}{k}.", co); sir(f"{b}{k + 1}.", co)
        conv(f"{b}{k + 2}.0.", co, co, 3); norm(f"{b}{k + 2}.1.", co); actp(f"{b}{k + 2}.2")
    for i, ci in enumerate(ins):
        embed(f"{e}t_embeddings.{i}.", ci)
    conv(e + "latent_layers.0.", cfg.latent_dim, st[-1], 3); norm(e + "latent_layers.1.", cfg.latent_dim); actp(e + "latent_layers.2")

    d = prefix + "decoder."
    rs = st[::-1]
    conv(d + "decoder_input_layers.0.", rs[0], cfg.latent_dim, 3); norm(d + "decoder_input_layers.1.", rs[0]); actp(d + "decoder_input_layers.2")
    outs = rs[1:] + [rs[-1]]
    for i, (ci, co) in enumerate(zip(rs, outs)):
        b = f"{d}layers.{i}.layers."
        conv(b + "0.0.", ci, 2 * ci, 3); norm(b + "0.1.", ci); actp(b + "0.2")
        k = 1
        if len(st) - 1 - i in cfg.attention_stages:
            out.update(conditional_shapes(b + "1.", ci, cfg.attention_depth, beta=has_beta))
            k = 2
        sir(f"{b}{k}.", ci)
        out[f"{b}{k + 1}.0.weight"] = (ci, co, 2, 2); out[f"{b}{k + 1}.0.bias"] = (co,); norm(f"{b}{k + 1}.1.", co); actp(f"{b}{k + 1}.2")
        sir(f"{b}{k + 2}.", co); sir(f"{b}{k + 3}.", co)
    for i, ci in enumerate(rs):
        embed(f"{d}t_embeddings.{i}.", ci)
    conv(d + "out_layers.0.", rs[-1], rs[-1], 3); actp(d + "out_layers.1")
    conv(d + "out_layers.2.", cfg.img_channels, rs[-1], 3)
    return out


def hash_params(cfg: UNetCfg, seed: int = 1, prefix: str = "model.", residual_gain: float = 0.3, head_gain: float = 1.0) -> Params:
    """Deterministic parameters for goldens: conv/linear weights ~ U(+-sqrt(3/fan_in)) (unit gain), biases
    ~ U(+-0.2), GroupNorm gamma ~ 1 +- 0.3, beta ~ U(+-0.3).  The GroupNorm that closes each inverted-residual
    branch is scaled by residual_gain, so blocks are perturbations of the identity as in a trained residual
    network; with gain 1 the 20 stacked blocks amplify any rounding difference by ~10x end to end, which makes
    whole-network comparisons across precisions uninformative.  head_gain scales the last convolution's weights:
    below 1 it keeps the Tanh head in its linear range, which makes the sampler's last steps (a fixed-point
    iteration x <- f(x)) contractive, as they are for a trained denoiser."""
    p: Params = {}
    for i, (k, shp) in enumerate(param_shapes(cfg, prefix).items()):
        if len(shp) >= 2:
            fan_in = 1
            for dd in shp[1:]:
                fan_in *= dd
            if len(shp) == 4 and shp[2] == 2:                                 # ConvTranspose2d (in,out,2,2): the only 2x2 kernels
                fan_in = shp[0]
            p[k] = hash_fill(shp, seed * 1000 + i, math.sqrt(3.0 / fan_in))
        elif k.endswith("weight"):
            p[k] = hash_fill(shp, seed * 1000 + i, 0.3, 1.0)
        elif k.endswith(".beta"):                                              # an activation's scalar: 1 +- 0.3
            p[k] = hash_fill((1,), seed * 1000 + i, 0.3, 1.0).reshape(())
        else:
            norm_bias = any(t in k for t in (".0.1.bias", ".3.1.bias", ".2.1.bias", "expansion.2.bias", "depthwise.2.bias",
                                              "projection.2.bias", "latent_layers.1.bias", "decoder_input_layers.1.bias"))
            p[k] = hash_fill(shp, seed * 1000 + i, 0.3 if norm_bias else 0.2)
        if "projection.2." in k:
            p[k] = p[k] * residual_gain
        if k.endswith("out_layers.2.weight"):
            p[k] = p[k] * head_gain
    return p


# --------------------------------------------------------------------------- autoencoder blocks (latent-diffusion configuration)
def batch_norm(p: Params, pre: str, x: Tensor, training: bool, momentum: float = 0.1, eps: float = 1e-5) -> Tensor:
    """torch.nn.BatchNorm2d.forward: batch statistics in training mode (the running statistics p[pre + 'running_*'] advance
    in place and the batch counter increments), the running statistics in eval mode."""
    if training and pre + "num_batches_tracked" in p:
        p[pre + "num_batches_tracked"] += 1
    return F.batch_norm(x, p[pre + "running_mean"], p[pre + "running_var"], p[pre + "weight"], p[pre + "bias"], training, momentum, eps)


def _conv_bn_act(p: Params, conv: str, bn: str, site: str, x: Tensor, act, training: bool, *, stride=1, padding=1, transposed=False) -> Tensor:
    if transposed:
        y = F.conv_transpose2d(x, _r(p[conv + "weight"]), None, stride=2)
    else:
        y = F.conv2d(x, _r(p[conv + "weight"]), None, stride=stride, padding=padding)
    return _r(act(batch_norm(p, bn, _r(y), training), site))


def ae_down_block(p: Params, pre: str, x: Tensor, act, drop: DropMasks, training: bool) -> Tensor:
    """autoencoders/constituent_blocks.py:56-88,100-114: conv3x3 s2 (no bias) + BatchNorm2d + act -> inverted residual ->
    conv3x3 (no bias) + BatchNorm2d + act"""
    z = _conv_bn_act(p, pre + "layers.0.", pre + "layers.1.", pre + "layers.2", x, act, training, stride=2)
    z = sir_block(p, pre + "layers.3.", z, 16, act, drop)
    return _conv_bn_act(p, pre + "layers.4.", pre + "layers.5.", pre + "layers.6", z, act, training)


def ae_up_block(p: Params, pre: str, x: Tensor, act, drop: DropMasks, training: bool) -> Tensor:
    """autoencoders/constituent_blocks.py:150-181,193-198: convT 2x2 s2 (no bias) + BatchNorm2d + act -> inverted residual ->
    conv3x3 (no bias) + BatchNorm2d + act"""
    z = _conv_bn_act(p, pre + "layers.0.", pre + "layers.1.", pre + "layers.2", x, act, training, transposed=True)
    z = sir_block(p, pre + "layers.3.", z, 16, act, drop)
    return _conv_bn_act(p, pre + "layers.4.", pre + "layers.5.", pre + "layers.6", z, act, training)


def ae_encoder(p: Params, x: Tensor, n_stages: int, act, drop: DropMasks, training: bool) -> Tensor:
    """base_ae.py:81-114: conv3x3 + act, then one DownBlock per stage"""
    z = _r(act(_r(F.conv2d(_r(x), _r(p["layers.0.0.weight"]), p["layers.0.0.bias"], padding=1)), "layers.0.1"))
    for i in range(n_stages):
        z = ae_down_block(p, f"layers.{i + 1}.", z, act, drop, training)
    return z


def ae_decoder(p: Params, z: Tensor, n_stages: int, act, drop: DropMasks, training: bool) -> Tensor:
    """base_ae.py:173-248: convT 2x2 s2 + act -> conv1x1 (no bias) + BatchNorm2d + act -> UpBlocks -> conv3x3 + act + conv3x3"""
    z = _r(act(_r(F.conv_transpose2d(_r(z), _r(p["layers.0.weight"]), p["layers.0.bias"], stride=2)), "layers.1"))
    z = _conv_bn_act(p, "layers.2.", "layers.3.", "layers.4", z, act, training, padding=0)
    for i in range(n_stages):
        z = ae_up_block(p, f"layers.{5 + i}.", z, act, drop, training)
    last = f"layers.{5 + n_stages}."
    z = _r(act(_r(F.conv2d(z, _r(p[last + "0.weight"]), p[last + "0.bias"], padding=1)), last + "1"))
    return F.conv2d(z, _r(p[last + "2.weight"]), p[last + "2.bias"], padding=1)


def hash_state(shapes: Dict[str, Tuple[int, ...]], seed: int) -> Params:
    """Deterministic values for a state_dict given as {key: shape} (parameters and BatchNorm buffers alike): weights of
    convolutions / linears ~ U(+-sqrt(3 / fan_in)), norm weights and activation scalars 1 +- 0.3, running variances 1 +- 0.3,
    biases and running means ~ U(+-0.2), batch counters 0."""
    p: Params = {}
    for i, (k, shp) in enumerate(shapes.items()):
        shp = tuple(shp)
        if k.endswith("num_batches_tracked"):
            p[k] = torch.zeros((), dtype=torch.long)
        elif len(shp) >= 2:
            fan_in = 1
            for dd in shp[1:]:
                fan_in *= dd
            if len(shp) == 4 and shp[2] == 2:                 # ConvTranspose2d (in, out, 2, 2)
                fan_in = shp[0]
            p[k] = hash_fill(shp, seed * 1000 + i, math.sqrt(3.0 / fan_in))
        elif k.endswith(".beta"):
            p[k] = hash_fill((1,), seed * 1000 + i, 0.3, 1.0).reshape(())
        elif k.endswith("running_var") or k.endswith("weight"):
            p[k] = hash_fill(shp, seed * 1000 + i, 0.3, 1.0)
        else:
            p[k] = hash_fill(shp, seed * 1000 + i, 0.2)
    return p
