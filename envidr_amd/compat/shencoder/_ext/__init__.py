from . import _shencoder               # noqa: F401
