from . import _gridencoder               # noqa: F401
