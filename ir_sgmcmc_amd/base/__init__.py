from .base_trainer import BaseTrainer  # noqa: F401
