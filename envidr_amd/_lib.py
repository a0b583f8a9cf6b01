"""ctypes binding of the C-ABI shared library (`libenvidr_amd.so`, include/envidr_amd.h).

There is deliberately NO fallback: if the HIP library is missing or an entry point is absent the
import / call raises.  A CPU fallback on the product path would void every parity claim.

`call("march_rays", ...)` accepts torch tensors (their `data_ptr()` is passed), Python numbers and
None (NULL); the current HIP stream of the tensors' device is appended automatically.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import torch  # imported first on purpose: maps torch's bundled libamdhip64.so.7 before our library

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("ENVIDR_AMD_LIB", _PKG / "libenvidr_amd.so"))

# One character per C argument (stream excluded):  p pointer | u uint32 | f float | i int
SIGNATURES: dict[str, str] = {
    # raymarching (include/envidr_amd.h, reference raymarching.h:7-18)
    "near_far_from_aabb": "pppufpp",
    "get_rays": "pffffuuupp",
    "sph_from_ray": "ppfup",
    "morton3D": "pup",
    "morton3D_invert": "pup",
    "packbits": "pufp",
    "get_scatter_idx": "pup",
    "march_rays_train": "pppffuuuuuupppppppp",
    "composite_rays_train_forward": "ppppuufuupppp",
    "composite_rays_train_backward": "ppppppppppuufppuu",
    "march_rays": "uuppppffuuuppppppp",
    "composite_rays": "uufuupppppppp",
    "compact_alive": "uppp",
    # hashencoder (hashencoder.h:13-15)
    "hash_encode_forward": "ppppuuuufuip",
    "hash_encode_backward": "pppppuuuufuipp",
    "hash_encode_second_backward": "ppppuuuufuipppp",
    # gridencoder (gridencoder.h:12-13)
    "grid_encode_forward": "ppppuuuufupui",
    "grid_encode_backward": "pppppuuuufuppui",
    # half-precision tables (the at::Half dispatch of the two grid encoders; include/envidr_amd.h "ABI 6")
    "hash_encode_forward_f16": "ppppuuuufuip",
    "hash_encode_backward_f16": "pppppuuuufuipp",
    "hash_encode_second_backward_f16": "ppppuuuufuipppp",
    "grid_encode_forward_f16": "ppppuuuufupui",
    "grid_encode_backward_f16": "pppppuuuufuppui",
    # freqencoder (freqencoder.h:6-9)
    "freq_encode_forward": "puuuup",
    "freq_encode_backward": "ppuuuup",
    # shencoder (shencoder.h:9-10)
    "sh_encode_forward": "ppuuup",
    "sh_encode_backward": "ppuuupp",
    "sh_encode_forward_f16": "ppuuup",
    "sh_encode_backward_f16": "ppuuupp",
    # ide_encoder (ide_encoder.py:98-130)
    "ide_encode_forward": "ppfuup",
    "ide_encode_backward": "pppfuupp",
}

_CTYPE = {"p": ctypes.c_void_p, "u": ctypes.c_uint32, "f": ctypes.c_float, "i": ctypes.c_int}


def argtypes(sig: str, with_stream: bool) -> list:
    t = [_CTYPE[c] for c in sig]
    if with_stream:
        t.append(ctypes.c_void_p)
    return t


class EnvidrError(RuntimeError):
    pass


_lib: ctypes.CDLL | None = None


def load() -> ctypes.CDLL:
    """Load the HIP library (once).  Raises if it has not been built: there is no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise EnvidrError(
            f"{LIB_PATH} not found: build it with `python -m envidr_amd.build` (hipcc, gfx950). "
            "envidr_amd has no CPU fallback by design.")
    lib = ctypes.CDLL(str(LIB_PATH), mode=ctypes.RTLD_GLOBAL)
    lib.envidr_last_error.restype = ctypes.c_char_p
    lib.envidr_abi_version.restype = ctypes.c_int
    for name, sig in SIGNATURES.items():
        fn = getattr(lib, "envidr_" + name, None)
        if fn is None:
            continue  # reported by exported_symbols(); calling it raises below
        fn.argtypes = argtypes(sig, with_stream=True)
        fn.restype = ctypes.c_int
    _lib = lib
    return lib


def release_scratch() -> int:
    """envidr_release_scratch(): hands back the buffers the library keeps between calls outside torch's allocator (the range-mask scratch of
    the table-gradient scatters); returns the bytes released.  Waits for the device."""
    lib = load()
    lib.envidr_release_scratch.restype = ctypes.c_uint64
    lib.envidr_release_scratch.argtypes = []
    return int(lib.envidr_release_scratch())


def exported_symbols() -> dict[str, bool]:
    lib = load()
    return {name: hasattr(lib, "envidr_" + name) for name in SIGNATURES}


_ELEMENT_DTYPES = {"float": (torch.float32,), "int32_t": (torch.int32,), "uint8_t": (torch.uint8, torch.bool),
                   "uint16_t": (torch.float16, torch.int16) + ((torch.uint16,) if hasattr(torch, "uint16") else ())}
def parse_header_types(header: Path | None = None) -> dict[str, list[str]]:
    """include/envidr_amd.h -> {operator: C element type of each argument ('' for scalars)}; the generator of _abi_types.py"""
    import re
    header = header or (_PKG.parent / "include" / "envidr_amd.h")
    text = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
    types = {}
    for m in re.finditer(r"\bint\s+envidr_([a-zA-Z0-9_]+)\s*\((.*?)\)\s*;", text, flags=re.S):
        params = [" ".join(p.split()) for p in m.group(2).split(",")][:-1]          # the stream is last
        types[m.group(1)] = [p.replace("const ", "").split("*")[0].strip() if "*" in p else "" for p in params]
    return types


def pointer_types() -> dict[str, list[str]]:
    """operator -> C element type of each of its arguments ('' for scalars): what the reference's CHECK_IS_FLOATING /
    CHECK_IS_INT guard (a tensor of another dtype would be read as garbage through a raw pointer).  The table ships inside the
    package (_abi_types.py, generated from include/envidr_amd.h), so the guard also works where the header is not installed."""
    from ._abi_types import ELEMENT_TYPES
    return ELEMENT_TYPES


def _ptr(x, name: str, pos: int):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        if not x.is_cuda:
            raise EnvidrError(f"{name}: argument {pos} must live on the GPU (got a {x.device} tensor)")
        if not x.is_contiguous():
            raise EnvidrError(f"{name}: argument {pos} must be contiguous")
        elem = pointer_types().get(name, [])
        if pos < len(elem) and elem[pos] in _ELEMENT_DTYPES and x.dtype not in _ELEMENT_DTYPES[elem[pos]]:
            raise EnvidrError(f"{name}: argument {pos} must be a {elem[pos]} tensor ({' / '.join(str(d) for d in _ELEMENT_DTYPES[elem[pos]])}), "
                              f"not {x.dtype}")
        return x.data_ptr()
    if isinstance(x, int):
        return x
    raise EnvidrError(f"{name}: argument {pos} must be a tensor, an address or None, not {type(x).__name__}")


def call(name: str, *args, stream: int | None = None) -> None:
    lib = load()
    sig = SIGNATURES[name]
    if len(args) != len(sig):
        raise EnvidrError(f"{name}: expected {len(sig)} arguments, got {len(args)}")
    fn = getattr(lib, "envidr_" + name, None)
    if fn is None:
        raise EnvidrError(f"libenvidr_amd.so does not export envidr_{name}")
    conv = []
    device = None
    for pos, (kind, a) in enumerate(zip(sig, args)):
        if kind == "p":
            if isinstance(a, torch.Tensor) and a.is_cuda and device is None:
                device = a.device
            conv.append(_ptr(a, name, pos))
        elif kind == "f":
            conv.append(float(a))
        else:
            conv.append(int(a))
    if stream is None:
        stream = torch.cuda.current_stream(device).cuda_stream if device is not None else 0
    rc = fn(*conv, stream)
    if rc != 0:
        raise EnvidrError(f"envidr_{name} failed ({rc}): {lib.envidr_last_error().decode()}")
