from . import ddpm, unets  # noqa: F401
