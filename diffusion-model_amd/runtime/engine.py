"""Host-side executor of the UNet denoiser on the HIP kernels.

The reference expresses the network as a tree of nn.Modules that autograd differentiates op by op
(unets.py:100-132,206-240; constituent_blocks.py:135-147,227-243; conv_block.py:132-136).  Here the same
arithmetic is a straight-line program of fused kernels driven from Python:

    conv (MFMA implicit GEMM; concat / time-embedding add / input dropout fused into its loader)
      -> GroupNorm + activation (+ output dropout mask, + residual) in one pass
    depthwise 3x3 stencil -> GroupNorm + activation

Every forward op records a closure on a tape; `backward` replays the tape in reverse.  Gradients of tensors
with two consumers (skip connections, residual inputs) are summed inside the later data-gradient kernel's
epilogue, so no separate accumulation pass exists.  Activations are NHWC bf16, statistics and parameter
gradients fp32.  The parameters themselves stay fp32 in torch layout; bf16 kernel-layout copies are cached
and refreshed when a parameter's version counter changes.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import os

import torch
from torch import Tensor

from . import hip, ops, pack
from .lds_program import LdsOutOfSpace, LdsProgram
from .params import GradStore, WeightCache, conv_dims

_LOADER_ACTS = (0, hip.ACT["gelu"], hip.ACT["silu"])   # activations the convolution loader can apply (normalise-on-load)

# state_dict entries that are buffers of a BatchNorm2d, not parameters: they travel with the parameters but have no gradient
BUFFER_SUFFIXES = (".running_mean", ".running_var", ".num_batches_tracked")

BF16 = torch.bfloat16
K33, K33S2, K11, KT22 = hip.CONV_3X3_S1, hip.CONV_3X3_S2, hip.CONV_1X1, hip.CONVT_2X2_S2


class Var:
    """An activation and (during backward) the gradient flowing into it."""
    __slots__ = ("v", "g", "need_grad")

    def __init__(self, v: Tensor, need_grad: bool = True):
        self.v = v
        self.g: Optional[Tensor] = None
        self.need_grad = need_grad


def attention_convs(convs: Dict[str, int], aliases: Dict[str, Tuple[str, int, int]], pre: str, c: int, kdim: int, depth: int) -> None:
    """Register the pointwise weights of a ConditionalModule (constituent_blocks.py:357-396) under `pre`.  With
    kdim == C nn.MultiheadAttention keeps one packed in_proj_weight [3C][C]: the logical names in_proj (all rows, the
    self-attention GEMM), in_proj_q and in_proj_kv (cross-attention: query rows / key+value rows) are row ranges of it.
    Otherwise it keeps q_proj_weight [C][C], k_proj_weight and v_proj_weight [C][kdim]."""
    for i in range(depth):
        m = f"{pre}mhas.{i}."
        if kdim == c:
            for name, r0, rows in (("in_proj", 0, 3 * c), ("in_proj_q", 0, c), ("in_proj_kv", c, 2 * c)):
                convs[m + name] = K11
                aliases[m + name + ".weight"] = (m + "in_proj_weight", r0, rows)
        else:
            for name in ("q_proj", "k_proj", "v_proj"):
                convs[m + name] = K11
                aliases[m + name + ".weight"] = (m + name + "_weight", 0, c)
        convs[m + "out_proj"] = K11
        convs[f"{pre}linears.{i}"] = K11


def _view_var(src: "Var", shape, tape) -> "Var":
    """A Var whose tensor is src's viewed as `shape`; in backward its gradient is handed to src (same memory, reshaped)."""
    out = Var(src.v.view(shape))

    def bwd():
        src.g = None if out.g is None else out.g.view(src.v.shape)
        out.g = None
    tape.append(bwd)
    return out


class UNetSpec:
    """Static description of the denoiser: which parameters exist, their kinds and shapes.
    Mirrors the constructor keywords the reference forwards to BaseUNet (dm_utils.py:39-58)."""

    def __init__(self, img_channels: int, stages_channels: Sequence[int], latent_dim: int, expand_factor: int = 2,
                 drop_p: float = 0.1, t_embed_dim: int = 32, activation: str = "gelu", base_groups_norm: int = 4,
                 time_norm: float = 100.0, attention_stages: Sequence[int] = (), attention_depth: int = 1):
        self.img_channels = int(img_channels)
        self.stages = [int(c) for c in stages_channels]
        self.latent_dim = int(latent_dim)
        self.expand = expand_factor
        self.drop_p = float(drop_p)
        self.t_dim = int(t_embed_dim)
        self.act = hip.ACT[activation]
        self.act_name = activation
        self.act_scalar = self.act in hip.ACT_WITH_SCALAR        # swish / hardswish: every activation module owns a learnable scalar
        self.groups = 4 * int(base_groups_norm)
        self.latent_groups = 8 * int(base_groups_norm)
        self.time_norm = float(time_norm)
        for c, g in [(c, self.groups) for c in self.stages] + [(self.latent_dim, self.latent_groups)]:
            if c % 32:
                raise ValueError(f"stage/latent channel counts must be multiples of 32 for the MFMA kernels, got {c}")
            cpg = c // g if g and c % g == 0 else 0
            if not (cpg in (1, 2, 4) or (cpg % 8 == 0 and 8 <= cpg <= 64)):
                raise ValueError(f"{c} channels in {g} normalisation groups: the GroupNorm kernels take 1, 2, 4 or a multiple of 8 (up to 64) "
                                 f"channels per group")
        if self.t_dim % 32:
            raise ValueError(f"t_embed_dim must be a multiple of 32, got {self.t_dim}")
        # stages whose Down/Up blocks carry a self-attention ConditionalModule at layers[1] (where the reference's
        # commented-out construction would insert it: constituent_blocks.py:92-101,194-203), and its depth
        self.attn_stages = sorted(int(i) for i in attention_stages)
        self.attn_depth = int(attention_depth)
        self.attn_heads = 8
        for i in self.attn_stages:
            if not 0 <= i < len(self.stages):
                raise ValueError(f"attention stage {i} out of range for {len(self.stages)} stages")
        # (name, kind) of every convolution weight, for packing
        self.convs: Dict[str, int] = {}
        self.dwconvs: List[str] = []
        self.aliases: Dict[str, Tuple[str, int, int]] = {}
        self._enumerate()

    def _sir(self, pre: str):
        self.convs[pre + "expansion.1"] = K11
        self.dwconvs.append(pre + "depthwise.1")
        self.convs[pre + "projection.1"] = K11

    def _enumerate(self):
        n = len(self.stages)
        e, d = "encoder.", "decoder."
        self.convs[e + "stem_layers.0"] = K33
        for i in range(n):
            b = f"{e}layers.{i}.layers."
            self.convs[b + "0.0"] = K33S2
            k = 1
            if i in self.attn_stages:
                attention_convs(self.convs, self.aliases, b + "1.", self.stages[i], self.stages[i], self.attn_depth)
                k = 2
            self._sir(f"{b}{k}.")
            self._sir(f"{b}{k + 1}.")
            self.convs[f"{b}{k + 2}.0"] = K33
            self.convs[f"{e}t_embeddings.{i}.layers.0"] = K11
            self.convs[f"{e}t_embeddings.{i}.layers.2"] = K11
        self.convs[e + "latent_layers.0"] = K33
        self.convs[d + "decoder_input_layers.0"] = K33
        for i in range(n):
            b = f"{d}layers.{i}.layers."
            self.convs[b + "0.0"] = K33
            k = 1
            if n - 1 - i in self.attn_stages:
                attention_convs(self.convs, self.aliases, b + "1.", self.stages[n - 1 - i], self.stages[n - 1 - i], self.attn_depth)
                k = 2
            self._sir(f"{b}{k}.")
            self.convs[f"{b}{k + 1}.0"] = KT22
            self._sir(f"{b}{k + 2}.")
            self._sir(f"{b}{k + 3}.")
            self.convs[f"{d}t_embeddings.{i}.layers.0"] = K11
            self.convs[f"{d}t_embeddings.{i}.layers.2"] = K11
        self.convs[d + "out_layers.0"] = K33
        self.convs[d + "out_layers.2"] = K33


class UNetEngine:
    """Runs BaseUNet.forward / its backward on the HIP kernels for one device."""

    def __init__(self, spec: UNetSpec):
        self.spec = spec
        self.wc = WeightCache(spec.convs, spec.dwconvs, getattr(spec, "aliases", None))
        self.grads: Optional[GradStore] = None
        self.seed: Optional[Tensor] = None
        self._mask_offsets: Dict[Tuple[int, int, int], Tuple[Tensor, List[int], int]] = {}
        self.mask_ahead = False        # set by a driver that draws keep bits one step ahead (it passes mask_ahead=True to forward)
        self._zero_arena: Optional[Tensor] = None
        self._zero_cursor = self._zero_need = 0
        self._mask_bufs: Dict = {}
        self._mask_key = None
        self._mask_seed = None
        self.fuse_embed = True          # all time-embedding MLPs in one forward and one backward launch
        self.wgrad_tiles_per_split = 8  # grouped weight gradients: 128-position tiles per position split
        self.wgrad_tiles_last = 4       # the same for the last phase of a multi-phase backward, whose group launch runs alone
        self.wgrad_tiles_by_phase = None
        self._cur_tps = 8
        self.group_wgrad = True         # defer the dense weight gradients to the end of backward and launch them together
        self._pin_eager: Optional[Tensor] = None      # pinned staging of the grouped-launch tables (eager steps reuse one)
        self._pin_event = None
        self._pin_spares: List[Tensor] = []           # handed one by one to graph captures, which then own them for good
        self._pin_captured: List[Tensor] = []
        self.fuse_sir = True            # one launch per inverted-residual block where the fused kernel supports the shape
        self.conv_stats = os.environ.get("DDPM_CONV_STATS", "1") != "0"   # large images: GroupNorm sums from the convolution's tiles
        self.fuse_scalar_act = os.environ.get("DDPM_FUSE_SCALAR_ACT", "1") != "0"   # swish / hardswish in the fused forward launches (A/B switch)
        self.dw_on_load = os.environ.get("DDPM_DW_ON_LOAD", "1") != "0"      # inference, large images: the expansion's GroupNorm + act in the depthwise loader
        self.norm_on_load = os.environ.get("DDPM_NORM_ON_LOAD", "1") != "0"  # inference, large images: GroupNorm + act in the consumer's loader
        # inference: the low-resolution stages as up to three launches with the activations resident in LDS (csrc/unet_lds.hip).
        # Off unless asked for (DDPM_LDS=1): at the benchmarked batch sizes the per-layer kernels are faster (a workgroup that
        # carries its images through every layer streams every weight itself, and the latency of that stream is not hidden)
        self.use_lds = os.environ.get("DDPM_LDS", "0") == "1"
        self.lds_ipw = int(os.environ.get("DDPM_LDS_IPW", "1"))      # images per workgroup of that launch ...
        self.lds_ipw_deep = int(os.environ.get("DDPM_LDS_IPW_DEEP", "8"))   # ... and of the launch that holds the deepest stages
        self._lds_s0: Dict[Tuple, Optional[int]] = {}
        self._lds_pin: Dict[str, list] = {}
        # reproducible gradients (parity runs): sums that the convolutions' epilogues would add with atomics (the per-image
        # sums feeding the time-embedding MLPs) are taken by a fixed-order kernel from the stored gradient instead, and the
        # library forms its scalar / column sums in one workgroup (ddpm_set_deterministic).  DDPM_DETERMINISTIC=1.
        self.deterministic = os.environ.get("DDPM_DETERMINISTIC", "0") == "1"
        self.always_repack = False      # set while capturing a HIP graph: parameter updates then happen without Python
        self.phases = 1                 # backward in 1, 2 (decoder | encoder), 4 (each halved at its middle stage), 5 (the last
                                        # of those cut again after the first encoder convolution) or, for any larger value,
                                        # one phase per stage, so
                                        # that a driver can run a phase's leaf work (weight gradients, exchange) beside the next phase
        self._reduce_tables: Dict[Tuple, Tuple[Tensor, List[int], List[int], int]] = {}
        self.external_zero = False      # gradient accumulation: backward passes add to the gradient sums, the driver clears them

    @property
    def bucketed(self) -> bool:
        return self.phases > 1

    @bucketed.setter
    def bucketed(self, v: bool) -> None:
        self.phases = 2 if v else 1

    # ------------------------------------------------------------------------------------------ helpers
    def _site_sizes(self, n: int, h: int, w: int) -> List[int]:
        """Element count of every dropout site in call order: for each inverted-residual block the block input,
        the expansion output and the depthwise output (conv_block.py:57-58,77-78,100-101)."""
        sp = self.spec
        sizes: List[int] = []

        def sir(c, hh, ww):
            e = int(sp.expand * c)
            sizes.extend([n * hh * ww * c, n * hh * ww * e, n * hh * ww * e])

        hh, ww = h, w
        for c in sp.stages:
            hh, ww = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
            sir(c, hh, ww); sir(c, hh, ww)
        rs = sp.stages[::-1]
        outs = rs[1:] + [rs[-1]]
        for ci, co in zip(rs, outs):
            sir(ci, hh, ww)
            hh, ww = 2 * hh, 2 * ww
            sir(co, hh, ww); sir(co, hh, ww)
        return sizes

    def _make_masks(self, key, sizes_fn: Callable[[], List[int]], device, ahead: bool = False) -> List[Tensor]:
        """One keep-bit buffer for all dropout sites of a pass (element counts from sizes_fn, cached under key).

        ahead = True (only the pipelined training step asks for it, per call): the buffer is persistent and is NOT filled
        here; the driver draws the next step's bits into a second buffer while this step computes (generate_masks) and
        copies them over once the step's last reader is done (commit_masks).  The first call fills the buffer directly.
        Every other caller (the module's autograd path, block programs, other shapes) gets freshly drawn bits, also on an
        engine a pipelined driver is attached to."""
        if key not in self._mask_offsets:
            sizes = sizes_fn()
            nbytes = [(s // 8 + 3) // 4 * 4 for s in sizes]
            offs = [0]
            for b in nbytes:
                offs.append(offs[-1] + b)
            self._mask_offsets[key] = (torch.tensor(offs, device=device, dtype=torch.long), offs, len(sizes))
        offs_dev, offs, nsites = self._mask_offsets[key]
        if self.seed is None:
            self.seed = torch.randint(0, 2 ** 62, (1,), device=device, dtype=torch.long)
        if ahead:
            if self._mask_key is not None and key != self._mask_key:
                raise RuntimeError(f"the pipelined step draws its dropout bits one step ahead for batch shape {self._mask_key[:3]}; "
                                   f"it was called with {key[:3]} (build a new step for a new shape)")
            if key not in self._mask_bufs:
                self._mask_bufs[key] = (torch.empty(offs[-1], device=device, dtype=torch.uint8), torch.empty(offs[-1], device=device, dtype=torch.uint8))
                self._mask_seed = self.seed ^ 0x5DEECE66D
                self._mask_key = key
                self.generate_masks(ahead=False)
                self.generate_masks(ahead=True)
            buf = self._mask_bufs[key][0]
        else:
            buf = torch.empty(offs[-1], device=device, dtype=torch.uint8)
            hip.call("ddpm_bump_seed", hip.ptr(self.seed), hip.stream())
            hip.call("ddpm_dropout_masks", hip.ptr(buf), hip.ptr(offs_dev), nsites, offs[-1], 1.0 - self.spec.drop_p, hip.ptr(self.seed),
                     hip.stream())
        return [buf[offs[i]:offs[i + 1]] for i in range(nsites)]

    def generate_masks(self, ahead: bool = True) -> None:
        """mask_ahead mode: draw fresh keep bits (own seed, advanced here) into the stand-by buffer (ahead) or the live one."""
        if self._mask_key is None:
            return
        offs_dev, offs, nsites = self._mask_offsets[self._mask_key]
        buf = self._mask_bufs[self._mask_key][1 if ahead else 0]
        hip.call("ddpm_bump_seed", hip.ptr(self._mask_seed), hip.stream())
        hip.call("ddpm_dropout_masks", hip.ptr(buf), hip.ptr(offs_dev), nsites, offs[-1], 1.0 - self.spec.drop_p, hip.ptr(self._mask_seed),
                 hip.stream())

    def commit_masks(self) -> None:
        """mask_ahead mode, after the last reader of a step: the stand-by bits become the live ones; the seed of the
        in-kernel (attention) dropout advances."""
        if self._mask_key is None:
            return
        live, ahead = self._mask_bufs[self._mask_key]
        live.copy_(ahead)
        hip.call("ddpm_bump_seed", hip.ptr(self.seed), hip.stream())

    def _open(self, params: Dict[str, Tensor], n: int, dev, rkey, train: bool, record: bool, masks, site_sizes, taps=None,
              mask_ahead: bool = False):
        """Start a pass: returns a namespace of the fused-unit builders (conv_unit, dw_unit, sir, embed, ...) bound to this
        pass's tape, dropout masks and partial-sum workspace.  rkey identifies the program + shape (workspace layout cache);
        site_sizes() lists the element counts of the dropout sites in call order."""
        sp = self.spec
        self.wc.refresh(params, dev, force=self.always_repack)
        if record:
            if self.grads is None or self.grads.flat.device != dev:
                self.grads = GradStore(sp.convs, sp.dwconvs, {k: v for k, v in params.items() if not k.endswith(BUFFER_SUFFIXES)}, dev,
                                       getattr(sp, "aliases", None))
        tape: Optional[List[Callable[[], None]]] = [] if record else None
        use_drop = train and sp.drop_p > 0.0
        if use_drop and masks is None:
            masks = self._make_masks(rkey, site_sizes, dev, mask_ahead)
        mscale = 1.0 / (1.0 - sp.drop_p) if use_drop else 1.0
        site = [0]
        P = params
        G = self.grads
        wc = self.wc

        # two-stage reductions: backward kernels write per-image / per-workgroup partial sums into one workspace,
        # a single pass at the end of backward folds them into the gradient buffer (no contended atomics)
        part_entries: List[Tuple[int, int, int, int, int]] = []
        part_cursor = [0]
        part_ws: List[Optional[Tensor]] = [None]

        pending: List = []                            # deferred weight-gradient problems: (descriptor, cost, tensors kept alive)
        late_wgrad: List[Callable[[], None]] = []     # deferred weight-gradient problems that run as launches of their own
        first_time: List[Tuple[Tensor, int]] = []     # (separately allocated piece, its offset) until the total is known

        def part_alloc(numel: int) -> Tuple[Tensor, int]:
            off = part_cursor[0]
            part_cursor[0] += (numel + 3) // 4 * 4
            if rkey not in self._reduce_tables:         # first backward for this shape: sizes are being discovered
                piece = torch.empty(numel, device=dev, dtype=torch.float32)
                first_time.append((piece, off))
                return piece, off
            if part_ws[0] is None:
                part_ws[0] = torch.empty(self._reduce_tables[rkey][3], device=dev, dtype=torch.float32)
            return part_ws[0][off:off + numel], off

        has_scalar = bool(getattr(sp, "act_scalar", False))

        def act_site(last: str) -> str:
            """state_dict prefix of the activation module that follows module `last` in its nn.Sequential ('x.1' -> 'x.2')"""
            head, _, idx = last.rpartition(".")
            return f"{head}.{int(idx) + 1}" if head else str(int(idx) + 1)

        def act_scalar(site: Optional[str], act: int) -> Optional[Tensor]:
            """the learnable scalar of a swish / hardswish module (None for the parameter-free activations)"""
            return P[site + ".beta"] if (has_scalar and act in hip.ACT_WITH_SCALAR) else None

        def scalar_partials(site: str, numel: int) -> Tensor:
            """workspace for the partial sums of an activation scalar's gradient; folded to one number with all the others"""
            buf, off = part_alloc(numel)
            part_entries.append((off, G.offset(site + ".beta"), numel, 0, 1))          # row stride 0 marks a scalar fold
            return buf

        def gn_partials(c: int, norm: Optional[str], conv_name: Optional[str]) -> Tensor:
            buf, off = part_alloc(n * 3 * c)
            if norm:
                part_entries.append((off, G.offset(norm + ".weight"), n, 3 * c, c))
                part_entries.append((off + c, G.offset(norm + ".bias"), n, 3 * c, c))
            if conv_name:
                part_entries.append((off + 2 * c, G.offset(conv_name + ".bias"), n, 3 * c, c))
            return buf

        def wgrad(kind, x0t, dyt, cout, name, x1=None, in_add=None, in_mask=None):
            """dense weight gradient: every position split stores its partial tile (no atomics on the few hot cache lines
            of a small weight block); the end-of-backward reduction folds the rows into the gradient buffer"""
            gw = G[name + ".weight"]
            tps = self._cur_tps if self.group_wgrad else 0
            if self.group_wgrad:
                # deep wide 3x3 problems at large batch leave the group for the pipelined kernel (a launch of their own, with
                # the group at the end of the phase)
                ps = ops.conv_wgrad_pipe_splits(kind, x0t, dyt, cout, x1=x1, in_add=in_add, in_mask=in_mask)
                if ps > 0:
                    numel = gw.numel()
                    buf, off = part_alloc(ps * numel)
                    late_wgrad.append(lambda: ops.conv_wgrad(kind, x0t, dyt, cout, None, x1=x1, in_add=in_add, in_mask=in_mask, in_scale=mscale,
                                                             part=buf))
                    dst = G.offset(name + ".weight")
                    for c0 in range(0, numel, 4096):
                        part_entries.append((off + c0, dst + c0, ps, numel, min(4096, numel - c0)))
                    return
            splits = ops.conv_wgrad_splits(kind, x0t, dyt, cout, x1=x1, tiles_per_split=tps)
            if splits <= 1 and not self.group_wgrad:
                ops.conv_wgrad(kind, x0t, dyt, cout, gw, x1=x1, in_add=in_add, in_mask=in_mask, in_scale=mscale)
                return
            if splits <= 1:              # single split: it adds straight into the gradient buffer (no contention)
                pending.append((ops.wgrad_desc(kind, x0t, dyt, cout, gw, None, x1, in_add, in_mask, mscale, tps),
                                ops.wgrad_cost(kind, x0t, dyt, cout, x1), (x0t, dyt, x1, in_add, in_mask, gw)))
                return
            numel = gw.numel()
            buf, off = part_alloc(splits * numel)
            if self.group_wgrad:
                pending.append((ops.wgrad_desc(kind, x0t, dyt, cout, None, buf, x1, in_add, in_mask, mscale, tps),
                                ops.wgrad_cost(kind, x0t, dyt, cout, x1), (x0t, dyt, x1, in_add, in_mask, buf)))
            else:
                ops.conv_wgrad(kind, x0t, dyt, cout, None, x1=x1, in_add=in_add, in_mask=in_mask, in_scale=mscale, part=buf)
            dst = G.offset(name + ".weight")
            for c0 in range(0, numel, 4096):
                part_entries.append((off + c0, dst + c0, splits, numel, min(4096, numel - c0)))

        def next_mask():
            if not use_drop:
                return None
            m = masks[site[0]]
            site[0] += 1
            return m

        def tap(name: str, v: Tensor):
            if taps is not None:
                taps[name] = v

        # ---- fused units ------------------------------------------------------------------------
        def conv_unit(name: str, x0: Var, *, x1: Optional[Var] = None, in_add: Optional[Var] = None, in_mask=None,
                      norm: Optional[str] = None, groups: int = 0, act: int = 0, res: Optional[Var] = None, out_mask=None,
                      c0_used: int = 0, in_affine=None) -> Var:
            """conv [-> GroupNorm] -> act [-> * out_mask] [+ res].  `norm` = parameter prefix of the GroupNorm.
            c0_used: only that many leading channels of x0 are non-zero (the zero-padded image at the stem).
            in_affine = (affine, act), inference only: x0 is raw, its GroupNorm + activation run in this convolution's loader."""
            kind = sp.convs[name]
            wf = wc.fwd[name]
            bias = P[name + ".bias"]
            cout = bias.shape[0]
            site = act_site(norm or name)
            bsc = act_scalar(site, act)           # swish / hardswish: in the GroupNorm epilogues (fused or per-unit), not in the bare
                                                  # convolution epilogues
            if bsc is None and norm is None and tape is None and res is None and out_mask is None:
                # inference, no normalisation: the activation runs in the convolution's epilogue (no pre-activation is kept)
                return Var(ops.conv_forward(kind, x0.v, wf, bias, cout, x1=None if x1 is None else x1.v,
                                            in_add=None if in_add is None else in_add.v, in_mask=in_mask, in_scale=mscale, act=act, c0_used=c0_used))
            gamma = P[norm + ".weight"] if norm else None
            beta = P[norm + ".bias"] if norm else None
            assert in_affine is None or (tape is None and norm and x1 is None and in_add is None)
            # deep, wide 3x3 convolutions at batch sizes that give every CU a 256-position tile: the matrix-core-bound kernel
            # plus a GroupNorm launch beats the whole-image kernels that finish the GroupNorm themselves (those are sized
            # for the latency-bound small batches)
            wide = ops.conv_pipe_wins(kind, x0.v, None if x1 is None else x1.v, cout) and in_mask is None and out_mask is None
            if (not wide and in_affine is None and (bsc is None or self.fuse_scalar_act) and norm and in_mask is None and out_mask is None
                    and ops.conv_gn_fusable(kind, x0.v, None if x1 is None else x1.v, cout, groups)):
                # low-resolution stages: a workgroup holds whole images and groups, the normalisation ends in the same launch
                # (swish / hardswish included: the epilogue reads the module's scalar)
                raw, y, stats = ops.conv_forward(kind, x0.v, wf, bias, cout, x1=None if x1 is None else x1.v,
                                                 in_add=None if in_add is None else in_add.v,
                                                 gn=(gamma, beta, groups, act, None if res is None else res.v, bsc), keep_raw=tape is not None)
            elif bsc is None and norm is None and res is None and out_mask is None and cout % 8 == 0 and x0.v.shape[1] > 4:
                # no normalisation: the activated tensor is a second result of the convolution's launch (the pre-activation
                # stays for the backward pass)
                raw, y = ops.conv_forward(kind, x0.v, wf, bias, cout, x1=None if x1 is None else x1.v,
                                          in_add=None if in_add is None else in_add.v, in_mask=in_mask, in_scale=mscale, act2=act,
                                          c0_used=c0_used)
                stats = None
            else:
                # large images: the convolution's tiles leave their sums for the normalisation (no statistics pass)
                want = groups if (norm and self.conv_stats and not c0_used) else 0
                raw = ops.conv_forward(kind, x0.v, wf, bias, cout, x1=None if x1 is None else x1.v,
                                       in_add=None if in_add is None else in_add.v, in_mask=in_mask, in_scale=mscale, c0_used=c0_used,
                                       gn_part_groups=want, in_affine=in_affine)
                raw, spart = raw if want else (raw, None)
                y, stats = ops.gn_act_forward(raw, gamma, beta, groups if norm else 0, act, res=None if res is None else res.v,
                                              mask=out_mask, mask_scale=mscale, act_beta=bsc, stat_part=spart)
            out = Var(y)
            if tape is not None:
                tape.append(conv_bwd(name, x0, x1, in_add, in_mask, raw, out, stats, norm, groups, act, res, out_mask))
            return out

        def conv_bwd(name, x0, x1, in_add, in_mask, raw, out, stats, norm, groups, act, res, out_mask):
            """backward of conv [-> GroupNorm] -> act [-> mask] [+ res] given what its forward saved"""
            kind = sp.convs[name]
            cout = P[name + ".bias"].shape[0]
            gamma = P[norm + ".weight"] if norm else None
            beta = P[norm + ".bias"] if norm else None
            site = act_site(norm or name)
            bsc = act_scalar(site, act)

            def bwd():
                dy = out.g
                if res is not None:                       # residual input receives the same gradient (no copy)
                    assert res.g is None
                    res.g = dy
                wd = wc.dgrad[name]
                c0 = x0.v.shape[3]
                dk = pack.DGRAD_KIND[kind]
                if (bsc is None and norm and out_mask is None and x0.need_grad
                        and ops.conv_gnb_fusable(dk, dy, c0, groups, masked=in_mask is not None)):
                    # the GroupNorm / activation backward runs in the loader of the data-gradient convolution (its tile
                    # holds whole images and whole groups), which also writes the raw gradient the weight-gradient launch
                    # reads later
                    draw = torch.empty_like(raw)
                    dsum = self._take_zeros(n * c0, dev).view(n, c0) if in_add is not None else None
                    if dsum is not None and self.deterministic:
                        # fixed-order sum of this convolution's own data-gradient, the other consumers' gradient added after it
                        own = ops.conv_forward(dk, dy, wd, None, c0, rows=c0,
                                               gnb=(raw, gamma, beta, groups, act, stats, gn_partials(cout, norm, name), draw))
                        ops.spatial_sum(own, dsum)
                        x0.g = own if x0.g is None else ops.gn_act_forward(own, None, None, 0, 0, res=x0.g)[0]
                    else:
                        x0.g = ops.conv_forward(dk, dy, wd, None, c0, res=x0.g, rows=c0, sum_out=dsum,
                                                gnb=(raw, gamma, beta, groups, act, stats, gn_partials(cout, norm, name), draw))
                    out.g = None
                    wgrad(kind, x0.v, draw, cout, name, x1=None if x1 is None else x1.v,
                          in_add=None if in_add is None else in_add.v, in_mask=in_mask)
                    if in_add is not None:
                        in_add.g = dsum
                    if x1 is not None:
                        c1 = x1.v.shape[3]
                        x1.g = ops.conv_forward(dk, draw, wd, None, c1, res=x1.g, row0=c0, rows=c1)
                    return
                splits = ops.act_backward_splits(raw) if (bsc is None and norm is None and out_mask is None) else 0
                if splits:
                    # bare activation: a streaming kernel with many workgroups per image; its partial bias-gradient rows
                    # are folded with all the others
                    buf, off = part_alloc(n * splits * cout)
                    part_entries.append((off, G.offset(name + ".bias"), n * splits, cout, cout))
                    draw = ops.act_backward(raw, dy, act, buf, splits)
                else:
                    draw = ops.gn_act_backward(raw, dy, gamma, beta, groups if norm else 0, act, stats,
                                               gn_partials(cout, norm, name), mask=out_mask, mask_scale=mscale, act_beta=bsc,
                                               dbeta_part=None if bsc is None else scalar_partials(site, n * cout))
                out.g = None
                wgrad(kind, x0.v, draw, cout, name, x1=None if x1 is None else x1.v,
                      in_add=None if in_add is None else in_add.v, in_mask=in_mask)
                if not x0.need_grad:
                    return                                  # network inputs (image, sinusoid) need no gradient
                # gradient of the broadcast add = per-image spatial sum of this conv's own data-gradient,
                # accumulated in fp32 inside the kernel (x0.g, added afterwards, belongs to other consumers)
                dsum = self._take_zeros(n * c0, dev).view(n, c0) if in_add is not None else None
                if dsum is not None and self.deterministic and in_mask is None:
                    own = ops.conv_forward(dk, draw, wd, None, c0, rows=c0)
                    ops.spatial_sum(own, dsum)
                    x0.g = own if x0.g is None else ops.gn_act_forward(own, None, None, 0, 0, res=x0.g)[0]
                else:
                    x0.g = ops.conv_forward(dk, draw, wd, None, c0, res=x0.g, out_mask=in_mask, out_scale=mscale, rows=c0, sum_out=dsum)
                if in_add is not None:
                    in_add.g = dsum
                if x1 is not None:
                    c1 = x1.v.shape[3]
                    x1.g = ops.conv_forward(dk, draw, wd, None, c1, res=x1.g, row0=c0, rows=c1)
            return bwd

        def conv_bn_unit(name: str, x0: Var, bn: str, act: int) -> Var:
            """bias-free conv -> BatchNorm2d -> act, the unit of the autoencoder blocks (autoencoders/constituent_blocks.py:58-68,
            77-86,151-161,170-179; base_ae.py:184-193 of the reference).  Training mode normalises with batch statistics and
            advances the module's running statistics; eval mode normalises with the running statistics."""
            kind = sp.convs[name]
            cout = conv_dims(kind, P[name + ".weight"])[0]
            raw = ops.conv_forward(kind, x0.v, wc.fwd[name], None, cout)
            site = act_site(bn)
            bsc = act_scalar(site, act)
            gamma, beta = P[bn + ".weight"], P[bn + ".bias"]
            y, stats = ops.batchnorm_act_forward(raw, gamma, beta, act, training=train, running_mean=P[bn + ".running_mean"],
                                                 running_var=P[bn + ".running_var"], num_batches=P.get(bn + ".num_batches_tracked"),
                                                 act_beta=bsc)
            out = Var(y)
            if tape is not None:
                def bwd():
                    buf, off = part_alloc(3 * cout)             # complete sums: the whole batch is one 'image' of the norm kernel
                    part_entries.append((off, G.offset(bn + ".weight"), 1, 3 * cout, cout))
                    part_entries.append((off + cout, G.offset(bn + ".bias"), 1, 3 * cout, cout))
                    draw = ops.batchnorm_act_backward(raw, out.g, gamma, beta, act, stats, buf, training=train, act_beta=bsc,
                                                      dbeta_part=None if bsc is None else scalar_partials(site, cout))
                    out.g = None
                    wgrad(kind, x0.v, draw, cout, name)
                    if x0.need_grad:
                        x0.g = ops.conv_forward(pack.DGRAD_KIND[kind], draw, wc.dgrad[name], None, x0.v.shape[3], res=x0.g)
                tape.append(bwd)
            return out

        def head_unit(name: str, x0: Var, act: int = 0) -> Var:
            """conv -> [tanh] -> NCHW fp32: a network's image head"""
            kind = sp.convs[name]
            bias = P.get(name + ".bias")
            cout = conv_dims(kind, P[name + ".weight"])[0]
            if act not in (0, hip.ACT["tanh"]):
                raise ValueError("head_unit: activation none or tanh")
            cpad = (cout + 15) // 16 * 16
            y = ops.conv_forward(kind, x0.v, wc.fwd[name], bias, cout, act=act, f32_nchw=True)
            out = Var(y)
            if tape is not None:
                def bwd():
                    n_, _, oh, ow = y.shape
                    dz = torch.empty(n_, oh, ow, cpad, device=dev, dtype=BF16)
                    dy = out.g.contiguous()
                    if act:
                        hip.call("ddpm_tanh_bwd_nchw_to_nhwc", hip.ptr(dy), hip.ptr(y), hip.ptr(dz), n_, cout, oh * ow, cpad, hip.stream())
                    else:
                        hip.call("ddpm_nchw_f32_to_nhwc_bf16", hip.ptr(dy), hip.ptr(dz), n_, cout, oh * ow, cpad, hip.stream())
                    if bias is not None:
                        ops.colsum(dz, G[name + ".bias"], cout)
                    wgrad(kind, x0.v, dz, cout, name)
                    if x0.need_grad:
                        x0.g = ops.conv_forward(pack.DGRAD_KIND[kind], dz, wc.dgrad[name], None, x0.v.shape[3], res=x0.g,
                                                c0_used=cout if cout <= 4 else 0)
                tape.append(bwd)
            return out

        def dw_unit(pre: str, xin: Var, out_mask, stride: int = 1) -> Var:
            """depthwise 3x3 -> GroupNorm -> act [-> * out_mask]   (conv_block.py:77-96).  stride 2 (conv_block.py:84; a form
            the reference's UNet never builds): with padding 1 the strided result is every second pixel of the stride-1
            result, so the stride-1 kernel runs and a strided copy thins it out (4x the arithmetic of a dedicated kernel, on
            a path nothing hot takes)."""
            name, norm = pre + "depthwise.1", pre + "depthwise.2"
            if stride == 2:
                raw, spart = ops.dwconv3x3(xin.v, wc.dw[name], P[name + ".bias"])[:, ::2, ::2, :].contiguous(), None
            else:
                raw, spart = ops.dwconv3x3(xin.v, wc.dw[name], P[name + ".bias"], gn_part_groups=sp.groups) if self.conv_stats else (
                    ops.dwconv3x3(xin.v, wc.dw[name], P[name + ".bias"]), None)
            gamma, beta = P[norm + ".weight"], P[norm + ".bias"]
            y, stats = ops.gn_act_forward(raw, gamma, beta, sp.groups, sp.act, mask=out_mask, mask_scale=mscale,
                                          act_beta=act_scalar(act_site(norm), sp.act), stat_part=spart)
            out = Var(y)
            if tape is not None:
                tape.append(dw_bwd(pre, xin, raw, out, stats, out_mask, stride))
            return out

        def dw_bwd(pre, xin, raw, out, stats, out_mask, stride: int = 1):
            name, norm = pre + "depthwise.1", pre + "depthwise.2"
            gamma, beta = P[norm + ".weight"], P[norm + ".bias"]

            def bwd():
                cch = raw.shape[3]
                bsc = act_scalar(act_site(norm), sp.act)
                draw = ops.gn_act_backward(raw, out.g, gamma, beta, sp.groups, sp.act, stats, gn_partials(cch, norm, None),
                                           mask=out_mask, mask_scale=mscale, act_beta=bsc,
                                           dbeta_part=None if bsc is None else scalar_partials(act_site(norm), n * cch))
                out.g = None
                if stride == 2:          # the adjoint of the thinning copy: the gradient on every second pixel, zeros between
                    wide = torch.zeros(xin.v.shape, device=draw.device, dtype=draw.dtype)
                    wide[:, ::2, ::2, :] = draw
                    draw = wide
                nb = ops.dwconv3x3_wgrad_blocks(*xin.v.shape)
                buf, off = part_alloc(nb * 10 * cch)
                if self.group_wgrad:     # leaf work: with the phase's other weight gradients, beside the next phase's chain
                    late_wgrad.append(lambda: ops.dwconv3x3_wgrad(xin.v, draw, buf))
                else:
                    ops.dwconv3x3_wgrad(xin.v, draw, buf)
                part_entries.append((off, G.offset(name + ".weight"), nb, 10 * cch, 9 * cch))
                part_entries.append((off + 9 * cch, G.offset(name + ".bias"), nb, 10 * cch, cch))
                assert xin.g is None
                xin.g = ops.dwconv3x3(draw, wc.dw_flip[name], None)
            return bwd

        def sir(pre: str, xin: Var, residual: bool = True, stride: int = 1) -> Var:
            """SeparableInvertResidual (conv_block.py:132-136).  The three nn.Dropout sites act on the block input
            (fused into the expansion conv's loader) and on the two intermediate activations (fused into the
            GroupNorm epilogue that produces them).  residual = False / stride = 2: the forms that change the shape and add
            no input (conv_block.py:51-54,84), on the per-unit kernels."""
            m_in, m_e, m_d = next_mask(), next_mask(), next_mask()
            _, hh, ww, cc = xin.v.shape
            ne, nd, npj = pre + "expansion.1", pre + "depthwise.1", pre + "projection.1"
            ech = P[nd + ".bias"].shape[0]
            if not residual or stride != 1:
                e = conv_unit(pre + "expansion.1", xin, in_mask=m_in, norm=pre + "expansion.2", groups=sp.groups, act=sp.act, out_mask=m_e)
                d = dw_unit(pre, e, m_d, stride)
                return conv_unit(pre + "projection.1", d, norm=pre + "projection.2", groups=sp.groups, act=0)
            if (self.fuse_sir and ops.sir_supported(cc, hh, ww, ech, sp.groups, sp.act)
                    and (self.fuse_scalar_act or sp.act not in hip.ACT_WITH_SCALAR)):
                # whole block in one launch (activations resident in LDS); training saves what the per-unit backward needs
                g1, g2, g3 = pre + "expansion.2", pre + "depthwise.2", pre + "projection.2"
                y, sv = ops.sir_forward(xin.v, wc.fwd[ne], P[ne + ".bias"], P[g1 + ".weight"], P[g1 + ".bias"], wc.dw[nd], P[nd + ".bias"],
                                        P[g2 + ".weight"], P[g2 + ".bias"], wc.fwd[npj], P[npj + ".bias"], P[g3 + ".weight"],
                                        P[g3 + ".bias"], sp.groups, sp.act, masks=(m_in, m_e, m_d), mask_scale=mscale,
                                        record=tape is not None,
                                        act_betas=(act_scalar(act_site(g1), sp.act), act_scalar(act_site(g2), sp.act)))
                out = Var(y)
                if tape is not None and not ops.sir_backward_supported(cc, hh, ww, ech, sp.groups, sp.act):
                    # forward fused, backward through the per-unit kernels (they read what the fused forward saved)
                    ev, dv = Var(sv["y_e"]), Var(sv["y_d"])
                    tape.append(conv_bwd(ne, xin, None, None, m_in, sv["raw_e"], ev, sv["st1"], g1, sp.groups, sp.act, None, m_e))
                    tape.append(dw_bwd(pre, ev, sv["raw_d"], dv, sv["st2"], m_d))
                    tape.append(conv_bwd(npj, dv, None, None, None, sv["raw_o"], out, sv["st3"], g3, sp.groups, 0, xin, None))
                elif tape is not None:
                    def bwd():
                        # data-gradient chain in one launch; the two pointwise weight gradients sum over the batch
                        # and stay separate GEMM launches
                        p1, p2, p3 = gn_partials(ech, g1, ne), gn_partials(ech, g2, None), gn_partials(cc, g3, npj)
                        nwg = ops.sir_workgroups(hh * ww, n)
                        dwbuf, off = part_alloc(nwg * 10 * ech)
                        part_entries.append((off, G.offset(nd + ".weight"), nwg, 10 * ech, 9 * ech))
                        part_entries.append((off + 9 * ech, G.offset(nd + ".bias"), nwg, 10 * ech, ech))
                        s1, s2 = act_site(g1), act_site(g2)
                        betas = (act_scalar(s1, sp.act), act_scalar(s2, sp.act))
                        dbs = (None, None) if betas[0] is None else (scalar_partials(s1, nwg * 8), scalar_partials(s2, nwg * 8))
                        dx, d_o, d_e = ops.sir_backward(out.g, sv, P[g1 + ".weight"], P[g1 + ".bias"], P[g2 + ".weight"], P[g2 + ".bias"],
                                                        P[g3 + ".weight"], wc.dgrad[npj], wc.dgrad[ne], wc.dw_flip[nd], sp.groups, sp.act,
                                                        p1, p2, p3, dwbuf, masks=(m_in, m_e, m_d), mask_scale=mscale,
                                                        act_betas=betas, dbeta_parts=dbs)
                        out.g = None
                        wgrad(K11, sv["y_d"], d_o, cc, npj)
                        wgrad(K11, xin.v, d_e, ech, ne, in_mask=m_in)
                        assert xin.g is None
                        xin.g = dx
                    tape.append(bwd)
                return out
            if (tape is None and m_in is None and m_e is None and m_d is None and self.conv_stats and self.norm_on_load and self.dw_on_load
                    and sp.act in _LOADER_ACTS and act_scalar(act_site(pre + "expansion.2"), sp.act) is None
                    and act_scalar(act_site(pre + "depthwise.2"), sp.act) is None):
                # inference on large images: BOTH inner GroupNorms run in their consumer's loader.  The expansion's result stays
                # raw as well: the depthwise kernel normalises each of its rows once on the way into an LDS ring
                g1, g2 = pre + "expansion.2", pre + "depthwise.2"
                raw_e = torch.empty(n, hh, ww, ech, device=dev, dtype=BF16)
                # (a workgroup of that kernel owns 16 image rows: it needs a batch that gives every CU two of them)
                if n * (hh // 16) >= 512 and ops.dwconv3x3_affine_supported(raw_e, sp.groups):
                    raw_e, spart1 = ops.conv_forward(K11, xin.v, wc.fwd[ne], P[ne + ".bias"], ech, out=raw_e, gn_part_groups=sp.groups)
                    if spart1 is not None:
                        _, aff1 = ops.gn_fold_affine(spart1, P[g1 + ".weight"], P[g1 + ".bias"], hh * ww)
                        raw_d, spart2 = ops.dwconv3x3_affine(raw_e, aff1, sp.act, wc.dw[nd], P[nd + ".bias"], sp.groups)
                        _, aff2 = ops.gn_fold_affine(spart2, P[g2 + ".weight"], P[g2 + ".bias"], hh * ww)
                        return conv_unit(pre + "projection.1", Var(raw_d), norm=pre + "projection.2", groups=sp.groups, act=0, res=xin,
                                         in_affine=(aff2, sp.act))
                    e = Var(ops.gn_act_forward(raw_e, P[g1 + ".weight"], P[g1 + ".bias"], sp.groups, sp.act)[0])
                else:
                    e = conv_unit(pre + "expansion.1", xin, norm=g1, groups=sp.groups, act=sp.act)
            else:
                e = conv_unit(pre + "expansion.1", xin, in_mask=m_in, norm=pre + "expansion.2", groups=sp.groups, act=sp.act, out_mask=m_e)
            if (tape is None and m_d is None and self.conv_stats and self.norm_on_load and sp.act in _LOADER_ACTS
                    and act_scalar(act_site(pre + "depthwise.2"), sp.act) is None):
                # inference on large images: the depthwise result stays raw; its GroupNorm + activation run in the loader of
                # the projection (one pass over the expanded tensor less)
                raw, spart = ops.dwconv3x3(e.v, wc.dw[nd], P[nd + ".bias"], gn_part_groups=sp.groups)
                if spart is not None:
                    g2 = pre + "depthwise.2"
                    _, affine = ops.gn_fold_affine(spart, P[g2 + ".weight"], P[g2 + ".bias"], hh * ww)
                    return conv_unit(pre + "projection.1", Var(raw), norm=pre + "projection.2", groups=sp.groups, act=0, res=xin,
                                     in_affine=(affine, sp.act))
                y, _ = ops.gn_act_forward(raw, P[pre + "depthwise.2.weight"], P[pre + "depthwise.2.bias"], sp.groups, sp.act)
                d = Var(y)
            else:
                d = dw_unit(pre, e, m_d)
            return conv_unit(pre + "projection.1", d, norm=pre + "projection.2", groups=sp.groups, act=0, res=xin)

        def embed(pre: str, sin: Var) -> Var:
            """EmbedBlock: Linear -> act -> Linear, as two 1x1 convolutions on a 1x1 'image' (constituent_blocks.py:308-328).
            Result: fp32 [N][C] (the layout the conv loader's broadcast add reads)."""
            h0 = conv_unit(pre + "layers.0", sin, act=sp.act)
            name = pre + "layers.2"
            bias = P[name + ".bias"]
            c = bias.shape[0]
            o = ops.conv_forward(K11, h0.v, wc.fwd[name], bias, c, f32_nchw=True)        # [N][C][1][1] fp32
            out = Var(o.view(n, c))
            if tape is not None:
                def bwd():
                    d = out.g.to(BF16).view(n, 1, 1, c)
                    ops.colsum(d, G[name + ".bias"], c)
                    wgrad(K11, h0.v, d, c, name)
                    h0.g = ops.conv_forward(K11, d, wc.dgrad[name], None, h0.v.shape[3])
                tape.append(bwd)
            return out

        attn_site = [0]

        def attn_module(pre: str, z: Var, cond: Optional[Var], depth: int, heads: int = 8) -> Var:
            """ConditionalModule.forward (constituent_blocks.py:402-437): depth x { multi-head attention of z over cond ->
            GroupNorm(1, C)(. + z) -> Linear -> act -> GroupNorm(1, C)(. + previous) }.  z: tokens [N, S, 1, C] (an NHWC
            activation viewed as tokens); cond: [N, Sk, 1, kdim], or None for self-attention (cond = the layer's own input).
            Projections and the feed-forward are pointwise GEMMs on the convolution kernel, the attention core and the two
            per-sample norms have their own kernels; attention-probability dropout is drawn in-kernel.
            The reference passes the same `condition` to every layer (:423-426), so with cond = None layer 0 is true
            self-attention (one packed q,k,v GEMM) and deeper layers attend over the module's input tokens."""
            c = z.v.shape[-1]
            s_q = z.v.numel() // (n * c)
            shape_in = z.v.shape
            cur = Var(z.v.view(n, s_q, 1, c), z.need_grad)
            first = cur
            if tape is not None:
                def hand_back():                        # runs last in backward: the token view's gradient is the input's
                    if first.g is not None:
                        assert z.g is None
                        z.g = first.g.view(shape_in)
                tape.append(hand_back)
            p_drop = sp.drop_p if use_drop else 0.0
            if p_drop > 0.0 and self.seed is None:
                self.seed = torch.randint(0, 2 ** 62, (1,), device=dev, dtype=torch.long)
            for i in range(depth):
                m, mn, lin, ln = f"{pre}mhas.{i}.", f"{pre}mha_norms.{i}", f"{pre}linears.{i}", f"{pre}linear_norms.{i}"
                packed = (m + "in_proj_weight") in P
                b_in = P[m + "in_proj_bias"]
                gb_in = G[m + "in_proj_bias"] if tape is not None else None
                sid = 0x1000 + attn_site[0]
                attn_site[0] += 1
                x_in = cur
                cnd = cond if cond is not None else (None if i == 0 else first)
                if cnd is None:
                    if not packed:
                        raise ValueError("self-attention needs kdim == vdim == in_channels (a packed in_proj_weight)")
                    qkv = ops.conv_forward(K11, x_in.v, wc.fwd[m + "in_proj"], b_in, 3 * c)
                    q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
                elif packed:
                    q = ops.conv_forward(K11, x_in.v, wc.fwd[m + "in_proj_q"], b_in[:c], c)
                    kv = ops.conv_forward(K11, cnd.v, wc.fwd[m + "in_proj_kv"], b_in[c:], 2 * c)
                    k, v = kv[..., :c], kv[..., c:]
                else:
                    q = ops.conv_forward(K11, x_in.v, wc.fwd[m + "q_proj"], b_in[:c], c)
                    k = ops.conv_forward(K11, cnd.v, wc.fwd[m + "k_proj"], b_in[c:2 * c], c)
                    v = ops.conv_forward(K11, cnd.v, wc.fwd[m + "v_proj"], b_in[2 * c:], c)
                o, lse = ops.attention_forward(q, k, v, heads, drop_p=p_drop, seed=self.seed, stream_id=sid, want_lse=tape is not None)
                r_raw = ops.conv_forward(K11, o, wc.fwd[m + "out_proj"], P[m + "out_proj.bias"], c, res=x_in.v)
                r, st1 = ops.sample_norm_forward(r_raw, None, 0, P[mn + ".weight"], P[mn + ".bias"])
                h_raw = ops.conv_forward(K11, r, wc.fwd[lin], P[lin + ".bias"], c)
                asite = f"{pre}activations.{i}"
                y, st2 = ops.sample_norm_forward(h_raw, r, sp.act, P[ln + ".weight"], P[ln + ".bias"], act_beta=act_scalar(asite, sp.act))
                out = Var(y)
                if tape is not None:
                    def bwd(m=m, mn=mn, lin=lin, ln=ln, packed=packed, cnd=cnd, gb_in=gb_in, sid=sid, x_in=x_in, q=q, k=k, v=v, o=o, lse=lse,
                            r_raw=r_raw, r=r, st1=st1, h_raw=h_raw, st2=st2, out=out, asite=asite):
                        bsc = act_scalar(asite, sp.act)
                        du2, da2 = ops.sample_norm_backward(h_raw, r, out.g, sp.act, P[ln + ".weight"], st2, gn_partials(c, ln, lin),
                                                            act_beta=bsc, dbeta_part=None if bsc is None else scalar_partials(asite, n))
                        out.g = None
                        wgrad(K11, r, da2, c, lin)
                        dr = ops.conv_forward(K11, da2, wc.dgrad[lin], None, c, res=du2)
                        du1, _ = ops.sample_norm_backward(r_raw, None, dr, 0, P[mn + ".weight"], st1, gn_partials(c, mn, m + "out_proj"))
                        wgrad(K11, o, du1, c, m + "out_proj")
                        d_o = ops.conv_forward(K11, du1, wc.dgrad[m + "out_proj"], None, c)
                        kw = dict(drop_p=p_drop, seed=self.seed, stream_id=sid)
                        sk = k.shape[1]
                        if cnd is None:
                            dqkv = torch.empty(n, s_q, 1, 3 * c, device=dev, dtype=BF16)
                            ops.attention_backward(q, k, v, o, lse, d_o, dqkv[..., :c], dqkv[..., c:2 * c], dqkv[..., 2 * c:], heads, **kw)
                            for j in range(3):
                                ops.colsum(dqkv[..., j * c:(j + 1) * c], gb_in[j * c:(j + 1) * c], c)
                            wgrad(K11, x_in.v, dqkv, 3 * c, m + "in_proj")
                            if x_in.need_grad:
                                gx = ops.conv_forward(K11, dqkv, wc.dgrad[m + "in_proj"], None, c, res=du1)
                                if x_in.g is not None:          # deeper layers already left their key/value gradients here
                                    gx, _ = ops.gn_act_forward(gx, None, None, 0, 0, res=x_in.g)
                                x_in.g = gx
                            return
                        dq = torch.empty(n, s_q, 1, c, device=dev, dtype=BF16)
                        if packed:
                            dkv = torch.empty(n, sk, 1, 2 * c, device=dev, dtype=BF16)
                            ops.attention_backward(q, k, v, o, lse, d_o, dq, dkv[..., :c], dkv[..., c:], heads, **kw)
                            ops.colsum(dq, gb_in[:c], c)
                            ops.colsum(dkv[..., :c], gb_in[c:2 * c], c)
                            ops.colsum(dkv[..., c:], gb_in[2 * c:], c)
                            wgrad(K11, x_in.v, dq, c, m + "in_proj_q")
                            wgrad(K11, cnd.v, dkv, 2 * c, m + "in_proj_kv")
                            if cnd.need_grad:
                                cnd.g = ops.conv_forward(K11, dkv, wc.dgrad[m + "in_proj_kv"], None, cnd.v.shape[-1], res=cnd.g)
                            qname = m + "in_proj_q"
                        else:
                            dk, dv = (torch.empty(n, sk, 1, c, device=dev, dtype=BF16) for _ in range(2))
                            ops.attention_backward(q, k, v, o, lse, d_o, dq, dk, dv, heads, **kw)
                            for t_, lo, nm in ((dq, 0, "q_proj"), (dk, c, "k_proj"), (dv, 2 * c, "v_proj")):
                                ops.colsum(t_, gb_in[lo:lo + c], c)
                                wgrad(K11, x_in.v if nm == "q_proj" else cnd.v, t_, c, m + nm)
                            if cnd.need_grad:
                                g_ = ops.conv_forward(K11, dk, wc.dgrad[m + "k_proj"], None, cnd.v.shape[-1], res=cnd.g)
                                cnd.g = ops.conv_forward(K11, dv, wc.dgrad[m + "v_proj"], None, cnd.v.shape[-1], res=g_)
                            qname = m + "q_proj"
                        if x_in.need_grad:
                            x_in.g = ops.conv_forward(K11, dq, wc.dgrad[qname], None, c, res=du1)
                    tape.append(bwd)
                cur = out
            return Var(cur.v.view(shape_in)) if tape is None else _view_var(cur, shape_in, tape)

        late: List[Tuple[int, Callable[[], None]]] = []   # (phase, backward work deferred to the flush of that phase)
        bounds: List[int] = []                        # len(part_entries) at every flush so far

        def flush(phase: int = 0, last: bool = True) -> bool:
            """End a phase of the backward pass: run the work deferred to it, launch the weight-gradient problems collected
            so far and fold the partial sums written so far into the gradient buffer.  Returns True when the gradients of
            this phase are final in the gradient buffer.  On the first backward pass of a shape the workspace layout is
            still being discovered, and everything is folded by the last flush."""
            for ph, fn in late:
                if ph == phase:
                    fn()
            for fn in late_wgrad:
                fn()
            late_wgrad.clear()
            if pending:
                self._launch_wgrad_group(pending, dev)
                pending.clear()
            bounds.append(len(part_entries))
            known = rkey in self._reduce_tables
            if not known:
                if not last:
                    return False
                if not part_entries:
                    return True                          # nothing was reduced in two stages in this pass
                tab = ops.reduce_table(part_entries, dev)
                widths = [max([e[4] for e in part_entries[lo:hi]] or [1]) for lo, hi in zip([0] + bounds[:-1], bounds)]
                self._reduce_tables[rkey] = (tab, list(bounds), widths, part_cursor[0])
                part_ws[0] = torch.empty(part_cursor[0], device=dev, dtype=torch.float32)
                for piece, off in first_time:
                    part_ws[0][off:off + piece.numel()].copy_(piece)
                ops.reduce_rows(part_ws[0], G.flat, tab, len(part_entries), max(widths))
                return True
            tab, tb, widths, _ = self._reduce_tables[rkey]
            k = len(bounds) - 1
            assert k < len(tb) and tb[k] == bounds[k], "the backward pass took a different path than the one its reduction table was built for"
            lo = tb[k - 1] if k else 0
            if tb[k] > lo:
                ops.reduce_rows(part_ws[0], G.flat, tab[lo:], tb[k] - lo, widths[k])
            return True

        from types import SimpleNamespace
        return SimpleNamespace(P=P, G=G, wc=wc, tape=tape, conv_unit=conv_unit, conv_bwd=conv_bwd, dw_unit=dw_unit, sir=sir, embed=embed,
                               attn_module=attn_module, conv_bn_unit=conv_bn_unit, head_unit=head_unit,
                               wgrad=wgrad, part_alloc=part_alloc, part_entries=part_entries, gn_partials=gn_partials, tap=tap, late=late,
                               finish=flush, mscale=mscale)

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, params: Dict[str, Tensor], x: Optional[Tensor], t: Tensor, train: bool, record: bool,
                masks: Optional[List[Tensor]] = None, taps: Optional[Dict[str, Tensor]] = None, mask_ahead: bool = False,
                head: Optional[Tuple[Tensor, Tensor]] = None):
        """x: NCHW fp32 on the GPU, t: int64 [N].  Returns (out NCHW fp32, tape or None).
        train enables dropout; record keeps what backward needs; mask_ahead: see _make_masks.
        head = (stem input as channel-padded NHWC bf16 [N,H,W,cpad], sinusoidal embedding bf16 [N,1,1,t_dim]) when a caller has
        produced both already (ddpm_train_head): x is then not read."""
        sp = self.spec
        t = t.to(torch.long).contiguous()
        if head is not None:
            n, h, w = head[0].shape[:3]
            c_img = sp.img_channels
            assert head[0].shape[3] == (c_img + 15) // 16 * 16 and head[0].dtype == BF16 and head[1].numel() == n * sp.t_dim
            x = head[0]
        else:
            assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == sp.img_channels
            x = x.contiguous()
            n, c_img, h, w = x.shape
        nst = len(sp.stages)
        if h % (1 << nst) or w % (1 << nst):
            raise ValueError(f"image size {h}x{w} must be divisible by {1 << nst}")
        dev = x.device
        c_ = self._open(params, n, dev, (n, h, w, self.phases), train, record, masks, lambda: self._site_sizes(n, h, w), taps, mask_ahead)
        P, G, wc, tape, late = c_.P, c_.G, c_.wc, c_.tape, c_.late
        conv_unit, sir, embed, wgrad, part_alloc, part_entries, tap, finish = (c_.conv_unit, c_.sir, c_.embed, c_.wgrad, c_.part_alloc,
                                                                               c_.part_entries, c_.tap, c_.finish)
        # ---- the network ------------------------------------------------------------------------
        cpad = (c_img + 15) // 16 * 16                 # the image (or latent) channels padded to the convolution kernels' input granule
        if head is not None:
            xin, sin = head[0], head[1].view(n, 1, 1, sp.t_dim)
        else:
            xin = torch.empty(n, h, w, cpad, device=dev, dtype=BF16)
            hip.call("ddpm_nchw_f32_to_nhwc_bf16", hip.ptr(x), hip.ptr(xin), n, c_img, h * w, cpad, hip.stream())
            sin = torch.empty(n, 1, 1, sp.t_dim, device=dev, dtype=BF16)
            hip.call("ddpm_time_embedding", hip.ptr(t), sp.time_norm, sp.t_dim, hip.ptr(sin), n, hip.stream())
        sin_v = Var(sin, need_grad=False)

        e, d = "encoder.", "decoder."
        # ---- time-embedding MLPs: they depend on t only, so all of them run here in one launch; their backward runs
        #      once at the end of the backward pass, when every stage-input gradient sum exists
        enames = [f"{e}t_embeddings.{i}." for i in range(nst)] + [f"{d}t_embeddings.{i}." for i in range(nst)]
        ewidths = [P[q + "layers.2.bias"].shape[0] for q in enames]
        temb_vars: Dict[str, Var] = {}
        embed_bwd_factory = None
        stage_seg: Dict[int, int] = {}                  # MLP index (encoder stages, then decoder stages) -> tape segment of its stage
        if (self.fuse_embed and not sp.act_scalar and sp.t_dim == 32 and len(enames) <= hip.EMBED_MAX and all(32 <= c <= 512 and c % 32 == 0 for c in ewidths)
                and all(P[q + "layers.0.bias"].shape[0] == 2 * c for q, c in zip(enames, ewidths))):
            outs, esaved = ops.embed_forward(sin, [wc.fwd[q + "layers.0"] for q in enames], [P[q + "layers.0.bias"] for q in enames],
                                             [wc.fwd[q + "layers.2"] for q in enames], [P[q + "layers.2.bias"] for q in enames], sp.act,
                                             record=tape is not None)
            for q, o in zip(enames, outs):
                temb_vars[q] = Var(o)
            if tape is not None:
                def bwd_embed(idx):
                    """backward of the MLPs enames[i], i in idx, in one launch"""
                    def run():
                        tiles = (n + 31) // 32
                        db1s, db0s = [], []
                        for i in idx:
                            q, c = enames[i], ewidths[i]
                            b1, o1 = part_alloc(tiles * c)
                            part_entries.append((o1, G.offset(q + "layers.2.bias"), tiles, c, c))
                            b0, o0 = part_alloc(tiles * 2 * c)
                            part_entries.append((o0, G.offset(q + "layers.0.bias"), tiles, 2 * c, 2 * c))
                            db1s.append(b1); db0s.append(b0)
                        dt16s, dzs = ops.embed_backward([temb_vars[enames[i]].g for i in idx], [esaved[i][0] for i in idx],
                                                        [wc.dgrad[enames[i] + "layers.2"] for i in idx], sp.act, db1s, db0s)
                        for i, d16, dz_ in zip(idx, dt16s, dzs):
                            wgrad(K11, esaved[i][1], d16, ewidths[i], enames[i] + "layers.2")
                            wgrad(K11, sin, dz_, 2 * ewidths[i], enames[i] + "layers.0")
                    return run
                embed_bwd_factory = bwd_embed
        z = conv_unit(e + "stem_layers.0", Var(xin, need_grad=False), act=sp.act, c0_used=c_img if c_img <= 4 else 0)
        tap("enc.stem", z.v)
        skips: List[Var] = []
        marks: List[int] = []                           # tape positions where a backward phase ends (forward order)
        cuts = self._phase_cuts(nst)

        def temb_of(pre: str, i: int) -> Var:
            return temb_vars.get(f"{pre}t_embeddings.{i}.") or embed(f"{pre}t_embeddings.{i}.", sin_v)

        # inference: everything from the first encoder stage whose images fit the LDS of a workgroup down to the latent and up
        # again to the same depth is ONE launch (the activations never leave the CU); the stages above it run layer by layer
        plan = self._lds_first_stage(P, wc, h, w, z.v.shape[3]) if (tape is None and taps is None) else None
        s0 = None if plan is None else plan[0]
        nenc = nst if s0 is None else s0
        for i in range(nenc):
            b = f"{e}layers.{i}.layers."
            if tape is not None and f"enc{i}" in cuts:
                marks.append(len(tape))
            stage_seg[i] = len(marks)
            temb = temb_of(e, i)
            a = conv_unit(b + "0.0", z, in_add=temb, norm=b + "0.1", groups=sp.groups, act=sp.act)
            if tape is not None and i == 0 and "enc0b" in cuts:
                marks.append(len(tape))
            k = 1
            if i in sp.attn_stages:
                a = c_.attn_module(b + "1.", a, None, sp.attn_depth, sp.attn_heads)
                k = 2
            a = sir(f"{b}{k}.", a)
            a = sir(f"{b}{k + 1}.", a)
            z = conv_unit(f"{b}{k + 2}.0", a, norm=f"{b}{k + 2}.1", groups=sp.groups, act=sp.act)
            skips.append(z)
            tap(f"enc.stage{i}", z.v)
        jfirst = 0
        if s0 is not None:
            jfirst = nst - 1 - s0                        # the decoder stage whose transposed-convolution unit ends the launch
            a = Var(self._lds_run(P, wc, z.v, lambda pre, i: temb_of(pre, i).v, n, plan))
        else:
            z = conv_unit(e + "latent_layers.0", z, norm=e + "latent_layers.1", groups=sp.latent_groups, act=sp.act)
            tap("enc.latent", z.v)
            if tape is not None and "dec" in cuts:          # everything recorded from here on is the decoder
                marks.append(len(tape))
            z = conv_unit(d + "decoder_input_layers.0", z, norm=d + "decoder_input_layers.1", groups=sp.groups, act=sp.act)
            tap("dec.input", z.v)
        for i in range(jfirst, nst):
            b = f"{d}layers.{i}.layers."
            k = 1
            if s0 is None or i > jfirst:
                if tape is not None and f"dec{i}" in cuts:
                    marks.append(len(tape))
                stage_seg[nst + i] = len(marks)
                temb = temb_of(d, i)
                a = conv_unit(b + "0.0", z, x1=skips[nst - 1 - i], in_add=temb, norm=b + "0.1", groups=sp.groups, act=sp.act)
                if nst - 1 - i in sp.attn_stages:
                    a = c_.attn_module(b + "1.", a, None, sp.attn_depth, sp.attn_heads)
                    k = 2
                a = sir(f"{b}{k}.", a)
                a = conv_unit(f"{b}{k + 1}.0", a, norm=f"{b}{k + 1}.1", groups=sp.groups, act=sp.act)
            a = sir(f"{b}{k + 2}.", a)
            z = sir(f"{b}{k + 3}.", a)
            tap(f"dec.stage{i}", z.v)
        hfeat = conv_unit(d + "out_layers.0", z, act=sp.act)
        name = d + "out_layers.2"
        bias = P[name + ".bias"]
        out = ops.conv_forward(K33, hfeat.v, wc.fwd[name], bias, c_img, act=hip.ACT["tanh"], f32_nchw=True)
        if tape is not None:
            out_holder = Var(out)
            head_dz: List[Optional[Tensor]] = [None]
            self._head_slot = (head_dz, G[name + ".bias"], (n, h, w, cpad, c_img))

            def bwd_head():
                if head_dz[0] is not None:
                    # a caller that fused the loss with the head's gradient (ddpm_elbo_tanh_head) brings the gradient of the
                    # logits and has added the bias gradient already
                    dz, head_dz[0] = head_dz[0], None
                else:
                    dz = torch.empty(n, h, w, cpad, device=dev, dtype=BF16)
                    dy = out_holder.g.contiguous()
                    hip.call("ddpm_tanh_bwd_nchw_to_nhwc", hip.ptr(dy), hip.ptr(out), hip.ptr(dz), n, c_img, h * w, cpad, hip.stream())
                    ops.colsum(dz, G[name + ".bias"], c_img)
                wgrad(K33, hfeat.v, dz, c_img, name)
                hfeat.g = ops.conv_forward(K33, dz, wc.dgrad[name], None, hfeat.v.shape[3], c0_used=c_img if c_img <= 4 else 0)
            tape.append(bwd_head)
            if embed_bwd_factory is not None:
                # an MLP's backward belongs to the phase that back-propagates its stage
                by_phase: Dict[int, List[int]] = {}
                for j, seg in stage_seg.items():
                    by_phase.setdefault(len(marks) - seg, []).append(j)
                for ph in sorted(by_phase):
                    late.append((ph, embed_bwd_factory(sorted(by_phase[ph]))))
            return out, (tape, out_holder, finish, marks)
        return out, None

    # ------------------------------------------------------------------------------------------ single blocks
    def run_program(self, params: Dict[str, Tensor], inputs: List[Tensor], needs_grad: List[bool], program, key, train: bool, record: bool,
                    site_sizes: Callable[[], List[int]]):
        """Run a block-level program (one of the reference's constituent blocks) built from the same fused units as the
        network.  inputs: device tensors in kernel layout (NHWC bf16, or bf16 [N,1,1,D] for an embedding input).
        program(units, input Vars) -> output Var.  Returns (output tensor, context for backward_program or None)."""
        dev = inputs[0].device
        c_ = self._open(params, inputs[0].shape[0], dev, key, train, record, None, site_sizes)
        in_vars = [Var(t, need_grad=record and g) for t, g in zip(inputs, needs_grad)]
        out = program(c_, in_vars)
        if not record:
            return out.v, None
        return out.v, (c_.tape, out, c_.finish, in_vars)

    def backward_program(self, ctx, dout: Tensor):
        """-> (gradients of the inputs (None where not needed), {parameter name: fp32 gradient in torch layout})"""
        tape, out, flush, in_vars = ctx
        self._cur_tps = self.wgrad_tiles_per_split
        self.grads.zero()
        out.g = dout
        for fn in reversed(tape):
            fn()
        flush(0, True)
        tape.clear()
        return [v.g for v in in_vars], self.grads.unpack()

    # ------------------------------------------------------------------------------------------ LDS-resident stages
    def _stage_table(self, arr, nbytes: int, dev, slot: str) -> Tensor:
        """Copy a host table (ctypes array) to a fresh device buffer through pinned memory on the launch stream.  Under graph
        capture the copy becomes a node that re-reads the pinned buffer at every replay, so that buffer is taken from the
        ones set aside by earlier eager passes and never reused afterwards."""
        import ctypes as C
        pinned = lambda: torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)                # noqa: E731
        st = self._lds_pin.setdefault(slot, [None, None])                                         # [eager buffer, its last copy's event]
        if torch.cuda.is_current_stream_capturing():
            fit = sorted((b.numel(), i) for i, b in enumerate(self._pin_spares) if b.numel() >= nbytes)
            if not fit:
                raise RuntimeError("no pinned staging buffer set aside for this capture: run the pass eagerly at least once before capturing it")
            host = self._pin_spares.pop(fit[0][1])
            self._pin_captured.append(host)
        else:
            if st[0] is None or st[0].numel() < nbytes:
                st[0] = pinned()
            elif st[1] is not None:
                st[1].synchronize()                    # the previous launch's copy must have read the buffer
            host = st[0]
            if len(self._pin_spares) < 64:
                self._pin_spares.append(pinned())
        C.memmove(host.data_ptr(), arr, nbytes)
        blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        blob.copy_(host[:nbytes], non_blocking=True)
        if not torch.cuda.is_current_stream_capturing():
            st[1] = torch.cuda.Event()
            st[1].record()
        return blob

    def _lds_build(self, P, wc, z: Tensor, temb: Callable[[str, int], Tensor], n: int, s0: int, s1: int, ipw_top: int, ipw_bot: int,
                   new: Callable[..., Tensor]):
        """The programs of the UNet between the input of encoder stage s0 (z: NHWC bf16) and the output of the transposed
        convolution unit of the decoder stage at the same depth (unets.py:127-131,230-237; constituent_blocks.py:135-147,
        227-243; conv_block.py:132-136 there), inference form.  Up to three launches: encoder stages [s0, s1) with ipw_top
        images per workgroup; stages [s1, last], the latent and decoder-input layers and the decoder up to the transposed
        convolution of the stage at depth s1 with ipw_bot (the deep stages hold nine tenths of the weights and a few KB of
        activations per image: a workgroup that streams them once for many images is what they want); the rest of the decoder
        with ipw_top again.  s1 == s0: one launch.  -> ([(program, cost bytes, flops)], output tensor)."""
        sp = self.spec
        nst, G, act = len(sp.stages), sp.groups, sp.act
        e, d = "encoder.", "decoder."
        rs = sp.stages[::-1]
        outs = rs[1:] + [rs[-1]]
        progs = []
        cur = {}

        def begin(ipw):
            cur["p"], cur["cost"] = LdsProgram(ipw, act), [0, 0.0]
            return cur["p"]

        def end():
            progs.append((cur["p"], cur["cost"][0], cur["cost"][1]))

        def conv(kind, src, dst, name, cin, cout, col0=0):
            wt = wc.fwd[name]
            cur["p"].conv(kind, src, dst, wt, P.get(name + ".bias"), cin, cout, col0)
            rows = n * (src.w * src.w if kind == KT22 else dst.w * dst.w)
            cur["cost"][0] += 2 * wt.numel()
            cur["cost"][1] += 2.0 * rows * cin * cout * (9 if kind in (K33, K33S2) else 4 if kind == KT22 else 1)

        def gn(t, norm, groups, a, ch, res=None):
            cur["p"].gn(t, P[norm + ".weight"], P[norm + ".bias"], groups, a, ch, 0, res)

        def load(t, g, col0=0):
            cur["p"].load(t, g, col0=col0)
            cur["cost"][0] += 2 * g.numel()

        def store(t, g):
            cur["p"].store(t, g)
            cur["cost"][0] += 2 * g.numel()

        def sir(x, pre, c, wide=0):
            prog = cur["p"]
            ech = P[pre + "depthwise.1.bias"].shape[0]
            et = prog.tile(x.w, ech)
            conv(K11, x, et, pre + "expansion.1", c, ech)
            gn(et, pre + "expansion.2", G, act, ech)
            prog.dw(et, wc.dw[pre + "depthwise.1"], P[pre + "depthwise.1.bias"], ech)
            gn(et, pre + "depthwise.2", G, act, ech)
            x2 = prog.tile(x.w, c + wide)
            conv(K11, et, x2, pre + "projection.1", ech, c)
            prog.free(et)
            gn(x2, pre + "projection.2", G, 0, c, res=x)
            prog.free(x)
            return x2

        skips = {}

        def enc_stage(tin, cin, i):
            prog = cur["p"]
            c, b = sp.stages[i], f"{e}layers.{i}.layers."
            prog.addt(tin, temb(e, i), cin)
            x = prog.tile(tin.w // 2, c)
            conv(K33S2, tin, x, b + "0.0", cin, c)
            prog.free(tin)
            gn(x, b + "0.1", G, act, c)
            x = sir(x, b + "1.", c)
            x = sir(x, b + "2.", c)
            y = prog.tile(x.w, c)
            conv(K33, x, y, b + "3.0", c, c)
            prog.free(x)
            gn(y, b + "3.1", G, act, c)
            skips[i] = new(n, y.w, y.w, c)
            store(y, skips[i])
            return y, c

        def dec_first_half(zt, j):
            """decoder stage j up to its transposed-convolution unit; zt: the stage input Z, in a tile wide enough for the
            skip connection behind it or (after an inverted-residual block, whose expanded tile leaves no room for a wide
            result) in a tile of its own that is copied into a wide one first"""
            prog = cur["p"]
            i, ci, co, b = nst - 1 - j, rs[j], outs[j], f"{d}layers.{j}.layers."
            if zt.ch < 2 * ci:
                wide = prog.tile(zt.w, 2 * ci)
                prog.mask(zt, wide, None, ci)
                prog.free(zt)
                zt = wide
            prog.addt(zt, temb(d, j), ci)
            load(zt, skips[i], col0=ci)
            a = prog.tile(zt.w, ci)
            conv(K33, zt, a, b + "0.0", 2 * ci, ci)
            prog.free(zt)
            gn(a, b + "0.1", G, act, ci)
            a = sir(a, b + "1.", ci)
            up = prog.tile(2 * a.w, co)
            conv(KT22, a, up, b + "2.0", ci, co)
            prog.free(a)
            gn(up, b + "2.1", G, act, co)
            return up

        def dec_second_half(up, j):
            co, b = outs[j], f"{d}layers.{j}.layers."
            a = sir(up, b + "3.", co)
            return sir(a, b + "4.", co)

        def hand_over(t, ch):
            g = new(n, t.w, t.w, ch)
            store(t, g)
            end()
            return g

        jlast = nst - 1 - s0                            # the decoder stage whose transposed-convolution unit ends the last launch
        jcut = nst - 1 - s1                             # ... the middle launch
        g, cin = z, z.shape[3]
        if s1 > s0:
            prog = begin(ipw_top)
            tin = prog.tile(g.shape[1], cin)
            load(tin, g)
            for i in range(s0, s1):
                tin, cin = enc_stage(tin, cin, i)
            end()                                       # (its last skip is the middle launch's input)
            g = skips[s1 - 1]
        prog = begin(ipw_bot)
        tin = prog.tile(g.shape[1], cin)
        load(tin, g)
        for i in range(s1, nst):
            tin, cin = enc_stage(tin, cin, i)
        L = sp.latent_dim
        y = prog.tile(tin.w, L)
        conv(K33, tin, y, e + "latent_layers.0", cin, L)
        prog.free(tin)
        gn(y, e + "latent_layers.1", sp.latent_groups, act, L)
        zt = prog.tile(y.w, 2 * rs[0])
        conv(K33, y, zt, d + "decoder_input_layers.0", L, rs[0])
        prog.free(y)
        gn(zt, d + "decoder_input_layers.1", G, act, rs[0])
        for j in range(jcut + 1):
            up = dec_first_half(zt, j)
            if j < jcut:
                zt = dec_second_half(up, j)
        out = hand_over(up, outs[jcut])
        if s1 > s0:
            prog = begin(ipw_top)
            up = prog.tile(out.shape[1], outs[jcut])
            load(up, out)
            zt = dec_second_half(up, jcut)
            for j in range(jcut + 1, jlast + 1):
                up = dec_first_half(zt, j)
                if j < jlast:
                    zt = dec_second_half(up, j)
            out = hand_over(up, outs[jlast])
        return progs, out

    def _lds_first_stage(self, P, wc, h: int, w: int, c_stem: int):
        """(s0, s1, images per workgroup of the deep launch): s0 = the first encoder stage from which the rest of the U fits
        the LDS of one workgroup, s1 >= s0 = the first stage that still fits with several images per workgroup.  None: no
        stage fits, or the network has a feature the LDS kernel lacks.  Found by laying the programs out for one workgroup's
        images on placeholder tensors; cached per input size."""
        sp = self.spec
        key = (h, w, self.lds_ipw, self.lds_ipw_deep)
        if key in self._lds_s0:
            return self._lds_s0[key]
        ok = (self.use_lds and not sp.attn_stages and not sp.act_scalar and sp.act in (hip.ACT["gelu"], hip.ACT["silu"], hip.ACT["relu"])
              and h == w and (h & (h - 1)) == 0 and sp.groups >= 8 and (sp.groups & (sp.groups - 1)) == 0
              and all((c & (c - 1)) == 0 for c in sp.stages + [sp.latent_dim]) and sp.expand == 2)
        found = None
        if ok:
            nst = len(sp.stages)
            cpu = lambda *shape: torch.empty(*shape, dtype=BF16)                                 # noqa: E731

            def fits(s0, s1, ipw_top, ipw_bot):
                cin = c_stem if s0 == 0 else sp.stages[s0 - 1]
                side = h >> s0
                if side >> (nst - s0) < 1 or side > 64:
                    return False
                nn = max(ipw_top, ipw_bot)
                try:
                    progs, _ = self._lds_build(P, wc, cpu(nn, side, side, cin),
                                               lambda pre, i: torch.empty(nn, P[f"{pre}t_embeddings.{i}.layers.2.bias"].shape[0]),
                                               nn, s0, s1, ipw_top, ipw_bot, cpu)
                    for prog, _, _ in progs:
                        prog.check()
                    return True
                except (LdsOutOfSpace, RuntimeError, AssertionError):
                    return False

            for s0 in range(nst):
                if fits(s0, s0, self.lds_ipw, self.lds_ipw):
                    found = (s0, s0, self.lds_ipw)
                    # the deep launch starts at the first stage of <= 4x4 pixels (where the weights outgrow the activations)
                    # or the first one after s0, with as many images per workgroup as fit
                    s1 = max(s0 + 1, next((i for i in range(nst) if (h >> (i + 1)) <= 4), nst))
                    deep = self.lds_ipw_deep
                    while s1 < nst and deep > self.lds_ipw:
                        if fits(s0, s1, self.lds_ipw, deep):
                            found = (s0, s1, deep)
                            break
                        deep //= 2
                    break
        self._lds_s0[key] = found
        return found

    def _lds_run(self, P, wc, z: Tensor, temb: Callable[[str, int], Tensor], n: int, plan) -> Tensor:
        dev = z.device
        s0, s1, deep = plan
        new = lambda *shape: torch.empty(*shape, device=dev, dtype=BF16)                        # noqa: E731
        progs, out = self._lds_build(P, wc, z, temb, n, s0, s1, self.lds_ipw, deep, new)
        for k, (prog, nbytes, flops) in enumerate(progs):
            arr, size = prog.table()
            blob = self._stage_table(arr, size, dev, f"lds{k}")
            ops._launch("unet_lds_kernel", lambda nbytes=nbytes, flops=flops, arr=arr, prog=prog: (nbytes, flops, f"{len(arr) - 1} ops, {prog.ipw} images per workgroup"),
                        lambda blob=blob, arr=arr, prog=prog: hip.call("ddpm_unet_lds_run", blob.data_ptr(), arr, len(arr), n, prog.ipw, prog.act, hip.stream()))
        return out

    def _launch_wgrad_group(self, pending, dev) -> None:
        """All deferred dense weight gradients in at most five launches.  The problem tables are built on the host,
        staged in pinned memory and copied to the device on the launch stream; under graph capture the copy becomes a
        graph node that re-reads the pinned buffer at every replay, so that buffer is never reused afterwards."""
        n = len(pending)
        arr = (hip.WgradDesc * n)(*[p[0] for p in pending])
        nbytes = int(hip.lib().ddpm_conv_wgrad_group_bytes(n))
        pinned = lambda: torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)                # noqa: E731
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            # pinned memory cannot be allocated during capture: take a buffer set aside by an earlier eager step
            fit = sorted((b.numel(), i) for i, b in enumerate(self._pin_spares) if b.numel() >= nbytes)
            fit = [i for _, i in fit]                  # smallest sufficient buffer first
            if not fit:
                raise RuntimeError("no pinned staging buffer set aside for this capture: run the step eagerly at least once before capturing it")
            host = self._pin_spares.pop(fit[0])
            self._pin_captured.append(host)
        else:
            if self._pin_eager is None or self._pin_eager.numel() < nbytes:
                self._pin_eager = pinned()
            elif self._pin_event is not None:
                self._pin_event.synchronize()          # the previous launch's copy must have read the buffer
            host = self._pin_eager
            if len(self._pin_spares) < 64:             # one more for a later capture of this launch (a few KB each)
                self._pin_spares.append(pinned())
        hip.call("ddpm_conv_wgrad_group_plan", arr, n, host.data_ptr(), nbytes)
        blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        blob.copy_(host[:nbytes], non_blocking=True)
        if not capturing:
            self._pin_event = torch.cuda.Event()
            self._pin_event.record()

        def cost():
            cs = [p[1]() for p in pending]
            return sum(c[0] for c in cs), sum(c[1] for c in cs), f"{n} problems"

        ops._launch("conv_wgrad_group_kernel", cost,
                    lambda: hip.call("ddpm_conv_wgrad_group_launch", host.data_ptr(), blob.data_ptr(), hip.stream()))

    # ------------------------------------------------------------------------------------------ backward
    def _phase_cuts(self, nst: int) -> set:
        """Where the tape is cut into backward phases: "enc{i}" / "dec{i}" = before encoder / decoder stage i (i >= 1), "dec" =
        before the decoder's input layers, "enc0b" = after the first convolution of encoder stage 0 (keeps the LAST phase,
        whose leaf work nothing can hide, down to that convolution and the stem)."""
        ph, mid = self.phases, nst // 2
        if ph <= 1:
            return set()
        if os.environ.get("DDPM_PHASE_CUTS"):
            return set(os.environ["DDPM_PHASE_CUTS"].split(","))
        if ph == 2:
            return {"dec"}
        if ph == 4:
            return {"enc1", "dec", f"dec{mid}"} if nst > 1 else {"dec"}
        if ph == 3:
            return {f"enc{mid}", "dec", f"dec{mid}"}
        if ph == 5:
            return {"enc0b", f"enc{mid}", "dec", f"dec{mid}"}
        return {"enc0b", "dec"} | {f"enc{i}" for i in range(1, nst)} | {f"dec{i}" for i in range(1, nst)}

    def phase_layout(self) -> Tuple[int, int]:
        """(number of backward phases, index of the phase after whose flush every `decoder.*` gradient is final)"""
        cuts = self._phase_cuts(len(self.spec.stages))
        return len(cuts) + 1, sum(1 for c in cuts if c.startswith("dec") and c != "dec")

    def backward(self, ctx, dout: Tensor, params: Dict[str, Tensor], unpack: bool = True) -> Optional[Dict[str, Tensor]]:
        """Replay the tape in reverse; returns {parameter name: fp32 gradient in torch layout}.  unpack=False (callers that
        capture the pass in a graph): the gradients are left in the gradient store's staging buffer (grads.flat_t)."""
        for ph in range(len(ctx[3]) + 1):
            self.backward_phase(ctx, ph, dout if ph == 0 else None)
        if not unpack:
            self.grads.unpack_staging()
            return None
        return self.grads.unpack()

    def backward_begin(self) -> None:
        """What opens a backward pass: the gradient sums and the pass's zeroed accumulators are cleared.  backward_data(phase 0)
        does it by itself unless told that the caller already has (begun=True)."""
        if not self.external_zero:
            self.grads.zero()
        if self._zero_arena is not None:
            self._zero_arena.zero_()
        self._zero_cursor = self._zero_need = 0

    def head_gradient_slot(self):
        """For a caller that computes the loss and the gradient of the last convolution's logits itself, after
        backward_begin() and before backward_data(ctx, 0, None, begun=True): (setter for the channel-padded NHWC bf16
        gradient of the logits, the fp32 bias gradient of the last convolution to add the gradient's per-channel sums to,
        (N, H, W, padded channels, image channels)) of the forward pass recorded last."""
        holder, bias_grad, dims = self._head_slot

        def put(dz: Tensor) -> None:
            holder[0] = dz
        return put, bias_grad, dims

    def backward_data(self, ctx, phase: int, dout: Optional[Tensor] = None, begun: bool = False) -> None:
        """The data-gradient chain of one backward phase (phase 0 starts from dout = gradient of the network output, or from
        the gradient of the logits handed over through head_gradient_slot()).
        Phases run head-first: with 2 phases, 0 = head + decoder and 1 = encoder; with 4, each of those is halved at its
        middle stage.  Leaf work (weight gradients, partial-sum folds, time-embedding MLPs) is only queued here."""
        tape, out_holder, _, marks = ctx
        self._cur_tps = self.wgrad_tiles_last if (marks and phase == len(marks)) else self.wgrad_tiles_per_split
        if self.wgrad_tiles_by_phase:
            self._cur_tps = self.wgrad_tiles_by_phase[min(phase, len(self.wgrad_tiles_by_phase) - 1)]
        if phase == 0:
            if not begun:
                self.backward_begin()
            out_holder.g = dout
        bounds = [0] + list(marks) + [len(tape)]
        seg = len(bounds) - 2 - phase
        for fn in reversed(tape[bounds[seg]:bounds[seg + 1]]):
            fn()

    def grads_zero(self) -> None:
        """Clear the gradient sums (external_zero mode: the driver decides when an accumulation window starts)."""
        if self.grads is not None:
            self.grads.zero()

    def backward_flush(self, ctx, phase: int) -> bool:
        """The leaf work of a phase; after it the gradients of the phase's parameters are final in the kernel-layout
        gradient buffer (returns False only on the very first pass of a shape, whose partial sums are all folded by the
        last phase's flush).  Touches nothing a later phase's data-gradient chain writes, so a driver may run it on a
        second stream beside that chain."""
        tape, _, flush, marks = ctx
        last = phase == len(marks)
        ready = flush(phase, last)
        if last:
            tape.clear()
            have = 0 if self._zero_arena is None else self._zero_arena.numel()
            if self._zero_need > have and not torch.cuda.is_current_stream_capturing():
                self._zero_arena = torch.empty(self._zero_need, device=self.grads.flat.device, dtype=torch.float32)
        return ready

    def _take_zeros(self, k: int, device) -> Tensor:
        """k zeroed floats for an accumulator of this backward pass.  They come from one arena cleared by a single fill at
        the start of the pass (sized by the previous pass; a fill kernel per accumulator costs a launch each: the
        time-embedding gradients alone need eight per step)."""
        k64 = (k + 63) // 64 * 64
        self._zero_need += k64
        a = self._zero_arena
        if a is not None and a.device == torch.device(device) and self._zero_cursor + k64 <= a.numel():
            out = a[self._zero_cursor:self._zero_cursor + k]
            self._zero_cursor += k64
            return out
        return torch.zeros(k, device=device, dtype=torch.float32)

    def backward_phase(self, ctx, phase: int, dout: Optional[Tensor] = None) -> bool:
        self.backward_data(ctx, phase, dout)
        return self.backward_flush(ctx, phase)
