"""Activation names of the reference registry (neural_nets/activations.py:102-114 there) mapped to the codes the
fused HIP kernels implement.  The two activations with a learnable scalar (swish, hardswish) and the rarely used
piecewise ones are not fused; asking for them is an error rather than a silent slow path."""
from __future__ import annotations

from ..runtime import hip

FUSED = {"gelu": hip.ACT["gelu"], "silu": hip.ACT["silu"], "relu": hip.ACT["relu"], "tanh": hip.ACT["tanh"],
         "sigmoid": hip.ACT["sigmoid"]}
KNOWN = ["sigmoid", "hardsigmoid", "tanh", "swish", "hardswish", "relu", "relu6", "leaky_relu", "selu", "gelu", "silu"]


def activation_code(name: str) -> int:
    if name not in KNOWN:
        raise KeyError(f"Activation must be one of these {KNOWN}, got {name} instead")
    if name not in FUSED:
        raise NotImplementedError(f"activation '{name}' has no fused HIP implementation; available: {sorted(FUSED)}")
    return FUSED[name]
