"""the native module of the reference's `gridencoder` extension (same function names and argument orders), on libenvidr_amd.so"""
from envidr_amd.compat.backends import EXTENSIONS as _E, make_backend as _make
_m = _make("gridencoder")
globals().update({n: getattr(_m, n) for n in _E["gridencoder"]})
__all__ = list(_E["gridencoder"])
