"""Module objects with the pybind surface of the reference's five CUDA extensions, generated from the C-ABI signature
table (envidr_amd._lib.SIGNATURES: one letter per argument, the same order as the reference's `*/src/*.h`)."""
from __future__ import annotations

import types

import torch

from .. import _lib

# extension -> functions it defines (reference */src/bindings.cpp)
EXTENSIONS = {
    "raymarching": ["packbits", "near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "get_scatter_idx",
                    "march_rays_train", "composite_rays_train_forward", "composite_rays_train_backward", "march_rays", "composite_rays"],
    "hashencoder": ["hash_encode_forward", "hash_encode_backward", "hash_encode_second_backward"],
    "gridencoder": ["grid_encode_forward", "grid_encode_backward"],
    "freqencoder": ["freq_encode_forward", "freq_encode_backward"],
    "shencoder": ["sh_encode_forward", "sh_encode_backward"],
}
_DOC = {"raymarching": "raymarching/src/bindings.cpp:5-19, raymarching.h:7-18", "hashencoder": "hashencoder/src/bindings.cpp:5-9, hashencoder.h:13-15",
        "gridencoder": "gridencoder/src/bindings.cpp:5-8, gridencoder.h:12-13", "freqencoder": "freqencoder/src/bindings.cpp:5-8, freqencoder.h:6-9",
        "shencoder": "shencoder/src/bindings.cpp:5-8, shencoder.h:9-10"}
# pointer arguments the reference declares at::optional<at::Tensor> (None allowed); everything else must be a tensor
_OPTIONAL = {"grid_encode_forward": {10}, "grid_encode_backward": {11, 12}, "sh_encode_forward": {5},
             # where the reference's wrappers pass a dummy 1-element tensor for "not wanted" (hashgrid.py:39-42, raymarching.py:268-271)
             # this library also accepts None
             "hash_encode_forward": {11}, "hash_encode_backward": {4, 12, 13}, "composite_rays_train_forward": {12}}
# the reference's hash_encode_* take a dummy 1-element dy_dx / grad_inputs when calc_grad_inputs is false (hashgrid.py:39-42)
_IGNORED_WHEN_OFF = {"hash_encode_forward": (10, [11]), "hash_encode_backward": (11, [12, 13])}
# AT_DISPATCH_FLOATING_TYPES_AND_HALF: the argument whose scalar type selects the instantiation (hashencoder.cu:747,778,817 inputs /
# grad / grad; gridencoder.cu:443,474 embeddings / grad) -> a half tensor there routes to the `_f16` entry point
_DISPATCH_ARG = {"hash_encode_forward": 0, "hash_encode_backward": 0, "hash_encode_second_backward": 0, "grid_encode_forward": 1, "grid_encode_backward": 0,
                 "sh_encode_forward": 0, "sh_encode_backward": 0}


def _make_fn(name: str):
    sig = _lib.SIGNATURES[name]
    optional = _OPTIONAL.get(name, set())
    gated = _IGNORED_WHEN_OFF.get(name)

    def fn(*args):
        if len(args) != len(sig):
            raise TypeError(f"{name}(): expected {len(sig)} arguments, got {len(args)}")
        conv = list(args)
        for pos, (kind, a) in enumerate(zip(sig, args)):
            if kind == "p":
                if a is None:
                    if pos not in optional:
                        raise RuntimeError(f"{name}(): argument {pos} must be a tensor, not None")
                    continue
                if not isinstance(a, torch.Tensor):
                    raise TypeError(f"{name}(): argument {pos} must be a torch.Tensor, not {type(a).__name__}")
                if not a.is_cuda:
                    raise RuntimeError(f"{name}(): argument {pos} must be a CUDA tensor")          # CHECK_CUDA
                if not a.is_contiguous():
                    raise RuntimeError(f"{name}(): argument {pos} must be a contiguous tensor")    # CHECK_CONTIGUOUS
            elif kind == "i":
                conv[pos] = int(bool(a)) if isinstance(a, bool) else int(a)
        if gated is not None and not conv[gated[0]]:
            for pos in gated[1]:
                conv[pos] = None
        target = name
        if name in _DISPATCH_ARG and conv[_DISPATCH_ARG[name]].dtype == torch.half:
            target = name + "_f16"
        elif name in _DISPATCH_ARG and conv[_DISPATCH_ARG[name]].dtype != torch.float32:
            raise RuntimeError(f"{name}(): floating tensors must be float32 or float16")           # CHECK_IS_FLOATING (no double here)
        try:
            _lib.call(target, *conv)
        except _lib.EnvidrError as e:
            raise RuntimeError(str(e)) from None

    fn.__name__ = name
    fn.__doc__ = f"{name}({', '.join('Tensor' if k == 'p' else {'u': 'int', 'f': 'float', 'i': 'bool/int'}[k] for k in sig)}) -> None   (HIP, gfx950)"
    return fn


_cache: dict[str, types.ModuleType] = {}


def make_backend(pkg: str) -> types.ModuleType:
    """the module the reference calls `_backend` for extension `pkg`"""
    if pkg not in EXTENSIONS:
        raise KeyError(f"unknown extension {pkg!r}; known: {sorted(EXTENSIONS)}")
    if pkg not in _cache:
        m = types.ModuleType(f"{pkg}._ext._{pkg}", f"libenvidr_amd.so behind the pybind surface of {_DOC[pkg]}")
        for name in EXTENSIONS[pkg]:
            setattr(m, name, _make_fn(name))
        _cache[pkg] = m
    return _cache[pkg]
