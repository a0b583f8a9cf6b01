"""Reference-shaped import surface: the five extension modules the reference's Python binds, under the reference's own
module names, backed by libenvidr_amd.so.

The reference's wrappers import their native code unconditionally as

    from raymarching._ext import _raymarching as _backend          (raymarching/raymarching.py:10)
    from hashencoder._ext import _hashencoder as _backend          (hashencoder/hashgrid.py:11-14)
    from gridencoder._ext import _gridencoder as _backend          (gridencoder/grid.py:9-13)
    from freqencoder._ext import _freqencoder as _backend          (freqencoder/freq.py:9-12)
    from shencoder._ext import _shencoder as _backend              (shencoder/sphere_harmonics.py:9-12)

and call `_backend.<fn>(tensors..., sizes...)` with the argument orders of `*/src/bindings.cpp` / `*/src/*.h`.
Two ways to give them this library instead of the CUDA extensions:

1. keep the reference's Python wrappers, replace only the native modules:

       import envidr_amd.compat as compat
       compat.install_backends()          # before the first `import raymarching` etc.

   registers `<pkg>._ext` and `<pkg>._ext._<pkg>` in sys.modules, so the imports above resolve to the shims
   (the same thing `build_ext.sh` achieves by moving the built .so into `<pkg>/_ext/`).

2. replace wrappers and native modules together: put this directory on sys.path ahead of the reference checkout,

       sys.path.insert(0, os.path.dirname(envidr_amd.compat.__file__))

   `import raymarching`, `hashencoder`, `gridencoder`, `freqencoder`, `shencoder`, `ide_encoder` then resolve to the
   packages here, which re-export this library's wrappers (`envidr_amd.raymarching` ...) and carry the `_ext` modules.

Every backend function takes torch tensors exactly like the pybind originals: outputs are caller-allocated and mutated
in place, `at::optional<Tensor>` arguments accept None, `bool` arguments accept Python bools, everything runs on the
current HIP stream of the tensors' device without synchronising, and the reference's TORCH_CHECKs (is-cuda, contiguous)
surface as RuntimeError.
"""
from __future__ import annotations

import sys
import types

from .backends import EXTENSIONS, make_backend

__all__ = ["install_backends", "EXTENSIONS", "make_backend"]


def install_backends(packages=None) -> dict:
    """register `<pkg>._ext` and `<pkg>._ext._<pkg>` for the reference's own wrappers (way 1 above); returns the modules"""
    done = {}
    for pkg in (packages or EXTENSIONS):
        backend = make_backend(pkg)
        ext_name = f"{pkg}._ext"
        ext = sys.modules.get(ext_name)
        if ext is None:
            ext = types.ModuleType(ext_name)
            ext.__path__ = []           # a package: `from pkg._ext import _pkg` looks the attribute up, then sys.modules
            sys.modules[ext_name] = ext
        setattr(ext, f"_{pkg}", backend)
        sys.modules[f"{ext_name}._{pkg}"] = backend
        parent = sys.modules.get(pkg)
        if parent is not None:
            setattr(parent, "_ext", ext)
        done[pkg] = backend
    return done
