from .adam_rate_decay import Adam  # noqa: F401
