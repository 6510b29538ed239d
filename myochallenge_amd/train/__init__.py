from .trainer import MyoTrainer  # noqa: F401
