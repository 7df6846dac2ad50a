from .optimizer import OPTIMIZERS, FusedAdam, load_checkpoint, save_checkpoint  # noqa: F401
