from . import _raymarching               # noqa: F401
