"""the native module of the reference's `raymarching` extension (same function names and argument orders), on libenvidr_amd.so"""
from envidr_amd.compat.backends import EXTENSIONS as _E, make_backend as _make
_m = _make("raymarching")
globals().update({n: getattr(_m, n) for n in _E["raymarching"]})
__all__ = list(_E["raymarching"])
