"""Activation names of the reference registry (neural_nets/activations.py:102-114 there) mapped to the codes the
HIP kernels implement.  All parameter-free ones are (gelu, silu and relu also inside the single-launch inverted-residual
kernels, the others through the per-unit kernels); the two with a learnable scalar (swish, hardswish) are not: asking for
them is an error rather than a silent slow path."""
from __future__ import annotations

from ..runtime import hip

FUSED = {k: hip.ACT[k] for k in ("gelu", "silu", "relu", "tanh", "sigmoid", "hardsigmoid", "relu6", "leaky_relu", "selu")}
KNOWN = ["sigmoid", "hardsigmoid", "tanh", "swish", "hardswish", "relu", "relu6", "leaky_relu", "selu", "gelu", "silu"]


def activation_code(name: str) -> int:
    if name not in KNOWN:
        raise KeyError(f"Activation must be one of these {KNOWN}, got {name} instead")
    if name not in FUSED:
        raise NotImplementedError(f"activation '{name}' has no fused HIP implementation; available: {sorted(FUSED)}")
    return FUSED[name]
