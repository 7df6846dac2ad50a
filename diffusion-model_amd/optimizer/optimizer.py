"""Optimizer registry with the reference's names (optimizer/optimizer.py:3-8 there) plus FusedAdam: Adam (and, when
asked, the trainer's EMA of the weights, base_trainer.py:102-106,278-282 there) as ONE HIP kernel over a flat buffer.

torch.optim.Adam on this network costs ~700 small kernels per step (350 tensors); the denoiser's whole
forward+backward is only a few hundred kernels, so the optimizer must not be a per-tensor loop either."""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
from torch import Tensor

from ..runtime import hip
from ..runtime.params import bump_weight_epoch, flat_grad_view


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (bias-corrected, L2 weight decay, no amsgrad) on a flat fp32 copy of the parameters.

    On the first step every parameter is re-pointed to a view of one flat buffer (values preserved), so state_dict(),
    .grad, DDP and EMA wrappers keep working on the same nn.Parameter objects.  The step counter and the learning
    rate live in device memory, so the step is capturable in a HIP graph.  A torch LR scheduler writes
    param_groups[0]["lr"] (the reference trainer always attaches CosineAnnealingLR, base_trainer.py:109-115 there): every
    eager step() compares that value with the one the device scalar holds and refreshes the scalar when it moved.  A
    captured step() cannot look at host values when it is replayed: call sync_lr() (or set_lr) after scheduler.step();
    the graph drivers in runtime/graphs.py do so before every replay.

    One hyper-parameter set for all parameters: several param groups are accepted only when their lr, betas, eps and
    weight_decay agree."""

    def __init__(self, params: Iterable[Tensor], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 ema_decay: Optional[float] = None, **_ignored):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.ema_decay = ema_decay
        self._flat = None
        self._lr_host = None
        self._check_groups()

    def _check_groups(self) -> None:
        g0 = self.param_groups[0]
        for g in self.param_groups[1:]:
            for k in ("lr", "betas", "eps", "weight_decay"):
                if g[k] != g0[k]:
                    raise ValueError(f"FusedAdam applies one hyper-parameter set to every parameter: param groups differ in {k!r} "
                                     f"({g0[k]} vs {g[k]})")

    def add_param_group(self, param_group) -> None:
        super().add_param_group(param_group)
        if len(self.param_groups) > 1:
            self._check_groups()
        if getattr(self, "_flat", None) is not None:
            raise RuntimeError("FusedAdam: parameters cannot be added after the first step (the flat buffer is already laid out)")

    # -------------------------------------------------------------------------------------------------
    def _params(self) -> List[Tensor]:
        return [p for g in self.param_groups for p in g["params"] if p.requires_grad]

    def _flatten(self) -> None:
        ps = self._params()
        dev = ps[0].device
        if not ps[0].is_cuda:
            raise RuntimeError("FusedAdam runs on the HIP kernels only: parameters must live on an MI355X (cuda) device")
        offs, total = [], 0
        for p in ps:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(total, device=dev, dtype=torch.float32)
        for p, o in zip(ps, offs):
            flat[o:o + p.numel()].copy_(p.detach().reshape(-1))
            p.data = flat[o:o + p.numel()].view(p.shape)
        self._flat, self._offs, self._total = flat, offs, total
        self._m = torch.zeros_like(flat)
        self._v = torch.zeros_like(flat)
        self._g = torch.zeros_like(flat)
        self._ema = flat.clone() if self.ema_decay is not None else None
        self._step = torch.zeros(1, device=dev, dtype=torch.long)
        self._lr_host = float(self.param_groups[0]["lr"])
        self._lr = torch.full((1,), self._lr_host, device=dev, dtype=torch.float32)

    def set_lr(self, lr: float) -> None:
        for g in self.param_groups:
            g["lr"] = lr
        self.sync_lr()

    def sync_lr(self) -> None:
        """Bring the device-resident learning rate up to date with param_groups (what an LR scheduler writes).  One host
        comparison; a fill kernel on the current stream only when the value moved.  Not callable inside a graph capture."""
        if self._flat is None:
            return
        if len(self.param_groups) > 1:
            self._check_groups()
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_host:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FusedAdam: the learning rate changed during a graph capture; call sync_lr() before capturing")
            self._lr.fill_(lr)
            self._lr_host = lr

    def ema_parameters(self) -> List[Tensor]:
        """Views of the EMA weights, in parameter order (None unless ema_decay was given)."""
        ps = self._params()
        return [self._ema[o:o + p.numel()].view(p.shape) for p, o in zip(ps, self._offs)]

    def _flat_grad(self, ps: List[Tensor]) -> Tensor:
        """The gradients as one flat tensor in this optimizer's layout: zero-copy when they already are views of such a
        buffer (what the HIP engine's backward produces), otherwise gathered with one foreach copy."""
        flat = flat_grad_view(ps, self._offs, self._total)
        if flat is not None:
            return flat
        dst = [self._g[o:o + p.numel()].view(p.shape) for p, o in zip(ps, self._offs)]
        src = [p.grad if p.grad is not None else torch.zeros_like(p) for p in ps]
        torch._foreach_copy_(dst, src)
        return self._g

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._flat is None:
            self._flatten()
        ps = self._params()
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        g = self._flat_grad(ps)
        grp = self.param_groups[0]
        b1, b2 = grp["betas"]
        hip.call("ddpm_adam_step", hip.ptr(self._flat), hip.ptr(g), hip.ptr(self._m), hip.ptr(self._v), hip.ptr(self._ema), self._total,
                 hip.ptr(self._lr), float(grp["lr"]), float(b1), float(b2), float(grp["eps"]), float(grp["weight_decay"]),
                 float(self.ema_decay if self.ema_decay is not None else 0.0), float(grad_scale), hip.ptr(self._step), hip.stream())
        bump_weight_epoch()               # the update happened outside torch's view: invalidate cached weight copies
        return loss

    @torch.no_grad()
    def step_ranges(self, ranges, last: bool, grad_scale: float = 1.0) -> None:
        """One optimizer step taken in pieces: the update of the element ranges [lo, hi) of the flat parameter buffer (multiples
        of 4, e.g. GradStore.unpack_phase / unpack_where ranges), for a driver whose gradients become final in parts.  The
        update is element-wise, so the pieces of one step together give exactly step()'s numbers; the pieces must be
        disjoint, cover every parameter once per step, and only the last call says last=True (it advances the step count)."""
        if self._flat is None:
            self._flatten()
        ps = self._params()
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        g = self._flat_grad(ps)
        grp = self.param_groups[0]
        b1, b2 = grp["betas"]
        ranges = list(ranges)
        for i, (lo, hi) in enumerate(ranges):
            if not (0 <= lo < hi <= self._total and lo % 4 == 0 and (hi - lo) % 4 == 0):
                raise ValueError(f"FusedAdam.step_ranges: bad range [{lo}, {hi}) of {self._total}")
            at = lambda t: 0 if t is None else t.data_ptr() + 4 * lo
            hip.call("ddpm_adam_step_range", at(self._flat), at(g), at(self._m), at(self._v), at(self._ema), hi - lo,
                     hip.ptr(self._lr), float(grp["lr"]), float(b1), float(b2), float(grp["eps"]), float(grp["weight_decay"]),
                     float(self.ema_decay if self.ema_decay is not None else 0.0), float(grad_scale), hip.ptr(self._step),
                     int(last and i == len(ranges) - 1), hip.stream())
        if last:
            if not ranges:
                raise ValueError("FusedAdam.step_ranges: the last piece of a step cannot be empty")
            bump_weight_epoch()


    # ---- checkpoint interchange ----------------------------------------------------------------------------------------
    # The reference saves optimizer.state_dict() of torch.optim.Adam into its checkpoints (base_trainer.py:118-130,285-294
    # there).  The flat moment buffers are exposed in exactly that schema (per-parameter 'step', 'exp_avg', 'exp_avg_sq'),
    # so a checkpoint written here resumes under torch.optim.Adam and one written there resumes here.
    def state_dict(self):
        sd = super().state_dict()
        if self._flat is None:
            return sd
        ps = self._params()
        step = self._step.to(torch.float32).reshape(()).clone()
        sd["state"] = {i: {"step": step.clone(), "exp_avg": self._m[o:o + p.numel()].view(p.shape).clone(),
                           "exp_avg_sq": self._v[o:o + p.numel()].view(p.shape).clone()}
                       for i, (p, o) in enumerate(zip(ps, self._offs))}
        if self._ema is not None:
            sd["ema"] = [e.clone() for e in self.ema_parameters()]
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        state = state_dict.get("state", {})
        ema = state_dict.get("ema")
        super().load_state_dict({"state": {}, "param_groups": state_dict["param_groups"]})
        if not state:
            return
        if self._flat is None:
            self._flatten()
        ps = self._params()
        steps = []
        for i, (p, o) in enumerate(zip(ps, self._offs)):
            st = state.get(i, state.get(str(i)))
            if st is None:
                continue
            self._m[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
            self._v[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
            steps.append(int(float(st["step"])))
        if steps:
            if len(set(steps)) != 1:
                raise ValueError("FusedAdam keeps one step counter: the checkpoint's parameters are at different steps")
            self._step.fill_(steps[0])
        self._lr_host = float(self.param_groups[0]["lr"])
        self._lr.fill_(self._lr_host)
        if ema is not None and self._ema is not None:
            for dst, src in zip(self.ema_parameters(), ema):
                dst.copy_(src)


def save_checkpoint(path: str, epoch: int, model, optimizer, lr_scheduler=None) -> None:
    """Write a checkpoint in the reference's schema (base_trainer.py:118-130 there)."""
    torch.save({"num_epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                "lr_scheduler_state_dict": lr_scheduler.state_dict() if lr_scheduler is not None else None}, path)


def load_checkpoint(path: str, model, optimizer=None, lr_scheduler=None, map_location=None) -> int:
    """Resume from a checkpoint in the reference's schema (the reference only ever saves; base_trainer.py:40 has an unused
    start_epoch).  Returns the stored epoch.  Loaded with weights_only=True: nothing in the file is executed."""
    ck = torch.load(path, map_location=map_location, weights_only=True)
    model.load_state_dict(ck["model_state_dict"])
    if optimizer is not None and ck.get("optimizer_state_dict") is not None:
        optimizer.load_state_dict(ck["optimizer_state_dict"])
    if lr_scheduler is not None and ck.get("lr_scheduler_state_dict") is not None:
        lr_scheduler.load_state_dict(ck["lr_scheduler_state_dict"])
    bump_weight_epoch()
    return int(ck["num_epoch"])


OPTIMIZERS = {
    "adam": torch.optim.Adam,
    "adamw": torch.optim.AdamW,
    "nadam": torch.optim.NAdam,
    "sgd": torch.optim.SGD,
    "fused_adam": FusedAdam,
}
