from . import _freqencoder               # noqa: F401
