from . import ELBO  # noqa: F401
