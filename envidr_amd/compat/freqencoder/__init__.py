"""`freqencoder` as the reference's code imports it (envidr_amd/compat/__init__.py, way 2): this library's wrappers + `_ext`."""
from envidr_amd.freqencoder import *      # noqa: F401,F403
from envidr_amd import freqencoder as _impl
from . import _ext                # noqa: F401
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
