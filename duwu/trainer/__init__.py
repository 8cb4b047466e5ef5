from .trainer import BaseTrainer, DMTrainer  # noqa: F401
