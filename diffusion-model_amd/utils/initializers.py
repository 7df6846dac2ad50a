"""Weight-initialiser registry with the reference's names (utils/initializers.py:22-43 there), so a config's
`initializer:` key keeps its meaning.  Values are torch.nn.init functions applied in place."""
from __future__ import annotations

from typing import Callable, Union

from torch import Tensor
from torch.nn import init

INITIALIZERS = {
    "glorot_normal": init.xavier_normal_,
    "glorot_uniform": init.xavier_uniform_,
    "he_normal": init.kaiming_normal_,
    "he_uniform": init.kaiming_uniform_,
    "normal": lambda t: init.normal_(t, 0.0, 0.01),
    "zeros": init.zeros_,
    "ones": init.ones_,
}


def get_initializer(name: Union[str, Callable[[Tensor], Tensor]]) -> Callable[[Tensor], Tensor]:
    if callable(name):
        return name
    if name not in INITIALIZERS:
        raise ValueError(f"Valid initializers are {list(INITIALIZERS)}", f"Got {name} instead.")
    return INITIALIZERS[name]
