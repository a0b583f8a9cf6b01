from . import _hashencoder               # noqa: F401
