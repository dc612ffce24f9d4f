from .pytorch import PytorchTrainer  # noqa: F401
