from .optimizer import OPTIMIZERS, FusedAdam  # noqa: F401
