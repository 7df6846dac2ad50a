"""ELBOLoss: per-timestep weighted reconstruction loss of the reference (losses/ELBO.py:12-76 there) as one fused
forward kernel (weighted error + reduction) and one fused backward kernel."""
from __future__ import annotations

from typing import Dict

import torch
from torch import Tensor, nn

from ..runtime import hip


class _ElboFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred: Tensor, target: Tensor, t: Tensor, table: Tensor, bce: bool, weighted: bool):
        pred_c, target_c = pred.float().contiguous(), target.float().contiguous()
        n = pred_c.shape[0]
        chw = pred_c.numel() // n
        loss = torch.empty(1, device=pred.device, dtype=torch.float32)
        hip.call("ddpm_elbo_fwd", hip.ptr(pred_c), hip.ptr(target_c), hip.ptr(t), hip.ptr(table), hip.ptr(loss), n, chw, int(bce),
                 int(weighted), hip.stream())
        ctx.save_for_backward(pred_c, target_c, t, table)
        ctx.flags = (n, chw, int(bce), int(weighted))
        return loss.view(())

    @staticmethod
    def backward(ctx, gloss: Tensor):
        pred, target, t, table = ctx.saved_tensors
        n, chw, bce, weighted = ctx.flags
        g = torch.empty_like(pred)
        gs = gloss.float().contiguous()
        hip.call("ddpm_elbo_bwd", hip.ptr(pred), hip.ptr(target), hip.ptr(t), hip.ptr(table), hip.ptr(gs), hip.ptr(g), n, chw, bce,
                 weighted, hip.stream())
        return g, None, None, None, None, None


class ELBOLoss(nn.Module):
    def __init__(self, reconstruction_method: str = "mse", use_elbo: bool = True, eps: float = 1e-4, *args, **kwargs) -> None:
        super().__init__()
        assert reconstruction_method in ["mse", "bce"], (
            f"reconstruction_method must be one of ['mse', 'bce'], got {reconstruction_method} instead.")
        self.eps = float(eps)
        self.use_elbo = use_elbo if isinstance(use_elbo, bool) else str(use_elbo).lower() not in ("false", "0", "no")
        self.bce = reconstruction_method == "bce"
        self._tables = {}

    def _table(self, ddpm, device) -> Tensor:
        """[T+1][8] table whose weight column follows ELBO.py:66-67 of the reference, from the schedule attributes
        of whatever DDPM object is passed (this package's or the reference's)."""
        key = (id(ddpm), str(device), self.eps)
        if key not in self._tables:
            T = ddpm.T
            t = torch.arange(1, T + 1)
            g = lambda name: getattr(ddpm, name).detach().to("cpu", torch.float32)
            vars_t = (g("sqrt_one_minus_alphas_bar")[t - 1] * g("bw_std")[t]) ** 2
            w = 0.5 * g("square_betas")[t] * g("alphas_bar")[t - 1] / (vars_t * g("square_one_minus_alphas_bar")[t] + self.eps)
            tab = torch.zeros(T + 1, 8)
            tab[1:, 5] = w.flatten()
            self._tables[key] = tab.to(device)
        return self._tables[key]

    def forward(self, ddpm: nn.Module, T: Tensor, X_pred: Tensor, X_input: Tensor, **kwargs) -> Dict[str, Tensor]:
        if not X_pred.is_cuda:
            raise RuntimeError("ELBOLoss runs on the HIP kernels only: tensors must live on an MI355X (cuda) device")
        core = ddpm.module if kwargs.get("is_ddp") is True else ddpm
        table = self._table(core, X_pred.device)
        t = T.to(torch.long).contiguous()
        return {"Loss": _ElboFn.apply(X_pred, X_input, t, table, self.bce, bool(self.use_elbo))}
