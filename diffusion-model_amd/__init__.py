"""MI355X-native DDPM hot path: HIP kernels behind the reference project's nn.Module surface.

The directory name contains a hyphen, so it is imported under the alias ``dm_amd``
(see dm_amd_bootstrap.py at the repository root).
"""
__all__ = ["runtime"]
