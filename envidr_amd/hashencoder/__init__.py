from .hashgrid import HashEncoder  # noqa: F401
