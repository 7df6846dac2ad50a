"""Activation names of the reference registry (neural_nets/activations.py:102-114 there) mapped to the codes the
HIP kernels implement: gelu, silu and relu also inside the single-launch inverted-residual kernels and the convolution
epilogues, the others through the per-unit GroupNorm/activation kernels.  swish and hardswish carry one learnable scalar
per activation module (activations.py:25-63 there: x * sigmoid(beta x), x * hardsigmoid(beta x)); their modules register
a 0-dim `beta` parameter under the reference's state_dict key and the kernels produce its gradient."""
from __future__ import annotations

from ..runtime import hip

FUSED = {k: hip.ACT[k] for k in ("gelu", "silu", "relu", "tanh", "sigmoid", "hardsigmoid", "relu6", "leaky_relu", "selu", "swish", "hardswish")}
KNOWN = ["sigmoid", "hardsigmoid", "tanh", "swish", "hardswish", "relu", "relu6", "leaky_relu", "selu", "gelu", "silu"]


def activation_code(name: str) -> int:
    if name not in KNOWN:
        raise KeyError(f"Activation must be one of these {KNOWN}, got {name} instead")
    if name not in FUSED:
        raise NotImplementedError(f"activation '{name}' has no fused HIP implementation; available: {sorted(FUSED)}")
    return FUSED[name]


def has_scalar(name: str) -> bool:
    """True for the activations whose module owns a learnable scalar `beta`"""
    return activation_code(name) in hip.ACT_WITH_SCALAR
