"""Autoencoder side of the latent-diffusion configuration (neural_nets/autoencoders/ of the reference): the BatchNorm-based
Down/Up blocks and the Encoder / Decoder stacks built from them, on the HIP kernels."""
from .base_ae import Decoder, Encoder  # noqa: F401
from .constituent_blocks import DownBlock, UpBlock  # noqa: F401
