"""DDPM wrapper: noise schedule, forward noising q, reverse step and the T-step sampler, behind the reference's
class surface (neural_nets/denoise_models/ddpm.py:29-165 there) with the elementwise work on the HIP kernels."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple, Type, Union

import os

import torch
from torch import Tensor, nn
from torch.nn.modules.utils import _pair

from ...runtime import hip


def _as_device(d):
    return torch.device("cuda", d) if isinstance(d, int) else torch.device(d)


class DDPM(nn.Module):
    def __init__(self, denoise_model: Type[nn.Module], T: int, beta: float = 0.25, **kwargs) -> None:
        super().__init__()
        self.img_channels = kwargs["img_channels"]
        self.T = T
        self.device = kwargs["device"]
        self.model = denoise_model(**kwargs)          # T and beta are consumed here and not forwarded (as in the reference)
        self.reconstruction_method = kwargs["reconstruction_method"]

        # Schedule tensors: plain attributes (not buffers), shape (T+1,1,1,1), built with the reference's
        # op order so values agree to the last bit on the same machine (ddpm.py:44-62 there).
        betas = torch.linspace(0.0, beta, steps=T + 1, device=self.device).view(T + 1, 1, 1, 1)
        alphas = 1.0 - betas
        alphas_bar = torch.cumprod(alphas, dim=0)
        one_minus = 1.0 - alphas_bar
        self.alphas_bar = alphas_bar
        self.sqrt_alphas_bar = torch.sqrt(alphas_bar)
        self.square_betas = betas ** 2
        self.one_minus_alphas_bar = one_minus
        self.square_one_minus_alphas_bar = one_minus ** 2
        self.sqrt_one_minus_alphas_bar = torch.sqrt(one_minus)
        self.fw_mean_coef = self.sqrt_alphas_bar
        self.fw_std = self.sqrt_one_minus_alphas_bar
        self.x_previous_coef = torch.sqrt(alphas) / one_minus
        self.x0_pred_coef = betas / one_minus
        self.bw_std = torch.sqrt(betas / one_minus)
        self._table: Optional[Tensor] = None
        self._seed: Optional[Tensor] = None

    # ------------------------------------------------------------------------------------------------
    def schedule_table(self, device) -> Tensor:
        """fp32 [T+1][8] per-timestep scalars for the kernels: sqrt(abar), sqrt(1-abar), A, B, C of the reverse
        step, ELBO weight.  Products are formed in the reference's association order."""
        if self._table is None or self._table.device != torch.device(device):
            T = self.T
            t = torch.arange(1, T + 1)
            g = lambda name: getattr(self, name).detach().to("cpu", torch.float32)
            tab = torch.zeros(T + 1, 8)
            # row 0: q(x0, t=0) = x0 as in the reference (fw_mean_coef[0] = 1, fw_std[0] = 0); the reverse step and the ELBO
            # weight are undefined there (nan / inf in the reference's tables) and stay 0
            tab[0, 0] = 1.0
            tab[1:, 0] = g("sqrt_alphas_bar")[t].flatten()
            tab[1:, 1] = g("sqrt_one_minus_alphas_bar")[t].flatten()
            tab[1:, 2] = (g("one_minus_alphas_bar")[t - 1] * g("x_previous_coef")[t]).flatten()
            tab[1:, 3] = (g("sqrt_alphas_bar")[t - 1] * g("x0_pred_coef")[t]).flatten()
            tab[1:, 4] = (g("sqrt_one_minus_alphas_bar")[t - 1] * g("bw_std")[t]).flatten()
            vars_t = (g("sqrt_one_minus_alphas_bar")[t - 1] * g("bw_std")[t]) ** 2
            tab[1:, 5] = (0.5 * g("square_betas")[t] * g("alphas_bar")[t - 1]
                          / (vars_t * g("square_one_minus_alphas_bar")[t] + 1e-4)).flatten()
            self._table = tab.to(device)
        return self._table

    def _rng_seed(self, device) -> Tensor:
        """Device-resident Philox seed, drawn once from torch's generator (so torch.manual_seed governs it) and
        advanced by a kernel after every use."""
        if self._seed is None or self._seed.device != torch.device(device):
            self._seed = torch.randint(0, 2 ** 62, (1,), device=device, dtype=torch.long)
        return self._seed

    @staticmethod
    def _need_gpu(x: Tensor, what: str):
        if not x.is_cuda:
            raise RuntimeError(f"DDPM.{what} runs on the HIP kernels only: tensors must live on an MI355X (cuda) device")

    def _check_t(self, t: Tensor, lo: int, what: str) -> Tensor:
        """The kernels index the per-timestep table with t; the reference's table look-up raises IndexError for a timestep
        outside the schedule.  Here a small kernel clamps t into [lo, T] (returned: what the caller's kernels read, so no
        launch ever indexes outside the table) and reports an out-of-range value through two words of pinned host memory.
        The host never waits for the device: the IndexError is raised by the next call that checks (q, reverse_q,
        check_timesteps), i.e. one call late when the device is still busy.  A synchronous check costs the drop-in training
        loop a full host/device rendezvous per micro-batch.  Skipped inside a graph capture, whose timesteps come from this
        module's own generators."""
        if torch.cuda.is_current_stream_capturing() or t.numel() == 0:
            return t
        self.check_timesteps()
        if getattr(self, "_t_flag", None) is None or not self._t_flag.is_pinned():      # (a deep copy of the module unpins it)
            self._t_flag = torch.zeros(2, dtype=torch.int32).pin_memory()
            self._t_what = ""
        safe = torch.empty_like(t)
        hip.call("ddpm_check_timesteps", hip.ptr(t), hip.ptr(safe), t.numel(), lo, self.T, self._t_flag.data_ptr(), hip.stream())
        self._t_what = f"DDPM.{what}: timesteps must lie in [{lo}, {self.T}]"
        if getattr(self, "strict_timesteps", False) or os.environ.get("DDPM_STRICT_TIMESTEPS") == "1":
            # the reference's behaviour to the letter: the offending call itself raises (ddpm.py:102-105,138-141 there), at the
            # price of one host/device rendezvous per call
            self.check_timesteps(wait=True)
        return safe

    def check_timesteps(self, wait: bool = False) -> None:
        """Raise the IndexError of an earlier q / reverse_q call that was handed a timestep outside the schedule (see _check_t);
        wait=True first waits for the device, so that every call made so far has been checked."""
        flag = getattr(self, "_t_flag", None)
        if flag is None:
            return
        if wait:
            torch.cuda.synchronize()
        if int(flag[0]) != 0:
            # (error path only) wait for the device before clearing the words: a checking kernel still in flight could
            # otherwise write its report between the read above and the clear, and the report would be lost
            torch.cuda.synchronize()
            bad = int(flag[1])
            flag.zero_()
            raise IndexError(f"{self._t_what}, got {bad}")

    # ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def q(self, x_0: Tensor, T: Tensor, noise: Optional[Tensor] = None) -> Tensor:
        """x_t = sqrt(abar_t) x_0 + sqrt(1 - abar_t) eps.  eps is drawn in-kernel unless `noise` is given."""
        self._need_gpu(x_0, "q")
        x_0 = x_0.float().contiguous()
        n, c, h, w = x_0.shape
        t = T.to(torch.long).contiguous()
        t = self._check_t(t, 0, "q")
        out = torch.empty_like(x_0)
        tab = self.schedule_table(x_0.device)
        if noise is not None:
            noise = noise.float().contiguous()
            hip.call("ddpm_q_sample", hip.ptr(x_0), hip.ptr(t), hip.ptr(noise), hip.ptr(tab), hip.ptr(out), None, 0, n, c, h * w,
                     None, 0, hip.stream())
        else:
            seed = self._rng_seed(x_0.device)
            hip.call("ddpm_q_sample", hip.ptr(x_0), hip.ptr(t), None, hip.ptr(tab), hip.ptr(out), None, 0, n, c, h * w,
                     hip.ptr(seed), 1, hip.stream())
            hip.call("ddpm_bump_seed", hip.ptr(seed), hip.stream())
        return out

    @torch.no_grad()
    def sample_timesteps(self, n: int, device=None, out: Optional[Tensor] = None) -> Tensor:
        """t ~ U{1..T} for n images (base_trainer.py:202 there: torch.randint(1, T + 1, (n,))), drawn in-kernel from the
        module's Philox seed (stream 3; the seed advances with the next q()).  Unlike torch.randint it involves no
        framework generator, so a captured training step carries no generator bookkeeping."""
        dev = _as_device(device if device is not None else self.device)
        if dev.type != "cuda":
            raise RuntimeError("DDPM.sample_timesteps runs on the HIP kernels only: it needs an MI355X (cuda) device")
        t = out if out is not None else torch.empty(n, device=dev, dtype=torch.long)
        hip.call("ddpm_sample_timesteps", hip.ptr(t), n, self.T, hip.ptr(self._rng_seed(dev)), 3, hip.stream())
        return t

    def forward(self, *args, **kwargs):
        return self.model(*args, **kwargs)

    @torch.no_grad()
    def reverse_q(self, x_t: Tensor, x0_predicted: Tensor, T: Tensor, noise: Optional[Tensor] = None,
                  out: Optional[Tensor] = None, _sigmoid: bool = False, _seed: Optional[Tensor] = None, _first_index: int = 0) -> Tensor:
        """x_{t-1} = A_t x_t + B_t x0_pred + C_t z (x0-form posterior).  z is drawn in-kernel (Philox) unless `noise` is
        given; _seed (a device int64 tensor, advanced by a kernel after use) replaces the module's own seed for callers
        that run several chains at once; _first_index: global flat index of x_t's first element in a job sharded over GPUs
        (the in-kernel noise is keyed by global element index)."""
        self._need_gpu(x_t, "reverse_q")
        x_t = x_t.float().contiguous()
        x0_predicted = x0_predicted.float().contiguous()
        t = T.to(torch.long).contiguous()
        t = self._check_t(t, 1, "reverse_q")
        n = x_t.shape[0]
        chw = x_t.numel() // n
        if out is None:
            out = torch.empty_like(x_t)
        tab = self.schedule_table(x_t.device)
        if noise is not None:
            noise = noise.float().contiguous()
            hip.call("ddpm_reverse_step", hip.ptr(x_t), hip.ptr(x0_predicted), hip.ptr(t), hip.ptr(noise), hip.ptr(tab), hip.ptr(out),
                     n, chw, None, 0, int(_sigmoid), 0, hip.stream())
        else:
            seed = _seed if _seed is not None else self._rng_seed(x_t.device)
            hip.call("ddpm_reverse_step", hip.ptr(x_t), hip.ptr(x0_predicted), hip.ptr(t), None, hip.ptr(tab), hip.ptr(out),
                     n, chw, hip.ptr(seed), 2, int(_sigmoid), int(_first_index), hip.stream())
            hip.call("ddpm_bump_seed", hip.ptr(seed), hip.stream())
        return out

    @torch.no_grad()
    def sample_tensor(self, N: int, shape: Union[int, Tuple[int, int]] = (64, 64), x_T: Optional[Tensor] = None,
                      noises: Optional[Sequence[Tensor]] = None, steps: Optional[int] = None) -> Tensor:
        """The sampler's loop (ddpm.py:153-162 there) returning the final tensor before the (x+1)/2 rescale.
        x_T / noises inject the start tensor and per-step noise (for parity checks); steps truncates the chain."""
        dev = _as_device(self.device)
        T = self.T
        times = torch.full((N,), T, device=dev, dtype=torch.long)
        x = x_T.to(dev).float() if x_T is not None else torch.randn((N, self.img_channels) + _pair(shape), device=dev)
        bce = self.reconstruction_method == "bce"
        k = 0
        while T >= 1 and (steps is None or k < steps):
            x0_pred = self.forward(x, times)
            x = self.reverse_q(x, x0_pred, times, noise=None if noises is None else noises[k].to(dev), _sigmoid=bce)
            T -= 1
            times -= 1
            k += 1
        self.check_timesteps(wait=True)       # the end of a chain is a natural rendezvous: nothing is reported late from here
        return x

    @torch.no_grad()
    def sample(self, N: int, shape: Union[int, Tuple[int, int]] = (64, 64)):
        """List of N PIL images, as the reference returns (ddpm.py:147-165, utils/other_utils.py:14-37 there)."""
        from PIL import Image
        arr = self.to_uint8(self.sample_tensor(N, shape)).permute(0, 2, 3, 1).cpu().numpy()
        return [Image.fromarray(a[..., 0] if a.shape[-1] == 1 else a) for a in arr]

    @staticmethod
    def to_uint8(x: Tensor) -> Tensor:
        """The byte image of a sampled tensor: (x + 1) / 2 clamped to [0, 1] (ddpm.py:163-165, utils/other_utils.py:14-37
        there), then torchvision's to_pil_image arithmetic, `pic.mul(255).byte()`: a truncation, not a rounding."""
        return ((x + 1) / 2).clamp_(0.0, 1.0).mul(255).to(torch.uint8)


class LatentDDPM(DDPM):
    """DDPM on autoencoder latents: the latent-diffusion configuration.  The reference only holds an unexecuted text of this
    class (ddpm.py:203-387 there: a triple-quoted string), whose intent is kept: the denoiser's "image" is the latent tensor
    (img_channels = latent channels), training data are encoder(images) with the autoencoder frozen and in eval mode, and
    samples are decoder(sampled latents).  Every part (denoiser, schedule, reverse step, Encoder, Decoder) is pinned on its
    own; no reference behaviour exists for the composition."""

    def __init__(self, denoise_model: Type[nn.Module], encoder: nn.Module, decoder: nn.Module, T: int, beta: float = 0.25, **kwargs) -> None:
        super().__init__(denoise_model, T, beta, **kwargs)
        self.encoder, self.decoder = encoder, decoder
        self.latent_channels = kwargs["img_channels"]
        for p in list(encoder.parameters()) + list(decoder.parameters()):
            p.requires_grad_(False)               # second-stage training moves the denoiser only

    @torch.no_grad()
    def encode(self, images: Tensor) -> Tensor:
        """images (N,C,H,W) -> latents (N, latent_channels, h, w), the x_0 of q() and of the training step"""
        self.encoder.eval()
        return self.encoder(images)

    @torch.no_grad()
    def decode(self, latents: Tensor) -> Tensor:
        self.decoder.eval()
        return self.decoder(latents)

    @torch.no_grad()
    def sample_images(self, N: int, latent_shape: Union[int, Tuple[int, int]]) -> Tensor:
        """the full T-step chain in latent space, then the decoder: (N, C, H, W) fp32 images"""
        return self.decode(self.sample_tensor(N, latent_shape))
