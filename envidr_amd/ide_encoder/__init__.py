from .ide_encoder import IntegratedDirEncoder  # noqa: F401
