from . import distributions, loss  # noqa: F401
