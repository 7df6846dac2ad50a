"""MI355X-native DDPM hot path: HIP kernels behind the reference project's nn.Module surface.

The directory name contains a hyphen, so it is imported under the alias ``dm_amd``
(see dm_amd_bootstrap.py at the repository root).  Module paths mirror the reference's:
``dm_amd.neural_nets.denoise_models.{ddpm,unets}``, ``dm_amd.losses.ELBO``.
"""
__all__ = ["runtime", "neural_nets", "losses", "optimizer", "utils"]
