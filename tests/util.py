"""Test helpers: run one C-ABI operator on any of the three implementations with numpy in/out.

backend = "oracle" (oracle/c restatement), "ref" (reference kernel bodies on CPU) or "hip"
(libenvidr_amd.so on cuda:0, through envidr_amd._lib -- i.e. through the C ABI).
Pointer arguments are numpy arrays (or None); they are copied, the call mutates the copies, and the
(possibly mutated) copies are returned in argument order so in-place outputs can be compared.
"""
from __future__ import annotations

import numpy as np

from envidr_amd._lib import SIGNATURES


def run_op(backend: str, name: str, *args):
    sig = SIGNATURES[name]
    assert len(sig) == len(args), (name, len(sig), len(args))
    if backend in ("oracle", "ref"):
        from oracle import clib
        lib = clib.oracle() if backend == "oracle" else clib.ref()
        work = [np.ascontiguousarray(a).copy() if (k == "p" and a is not None) else a for k, a in zip(sig, args)]
        lib.call(name, *work)
        return [w for k, w in zip(sig, work) if k == "p"]
    assert backend in ("hip", "shim", "refhip", "refhip_exact")
    import torch
    from envidr_amd import _lib
    dev = torch.device("cuda:0")
    work = [torch.from_numpy(np.ascontiguousarray(a).copy()).to(dev) if (k == "p" and a is not None) else a
            for k, a in zip(sig, args)]
    if backend in ("refhip", "refhip_exact"):
        # the reference's own kernels, compiled by hipcc, on the same device arrays (oracle/ref/device_keywords.h)
        from oracle import clib
        torch.cuda.synchronize()
        (clib.ref_hip() if backend == "refhip" else clib.ref_hip_exact()).call(name, *[(w.data_ptr() if isinstance(w, torch.Tensor) else w) for w in work])
        torch.cuda.synchronize()
        return [None if w is None else w.cpu().numpy() for k, w in zip(sig, work) if k == "p"]
    if backend == "shim":
        # through the reference-named backend modules (envidr_amd.compat): `<pkg>._ext._<pkg>.<name>(tensors...)`
        from envidr_amd.compat.backends import EXTENSIONS, make_backend
        base = name[:-4] if name.endswith("_f16") else name          # half cases go through the SAME pybind name, with half tensors
        pkg = next(p for p, names in EXTENSIONS.items() if base in names)
        targs = [bool(a) if k == "i" else a for k, a in zip(sig, work)]
        if base != name:
            targs = [a.view(torch.float16) if isinstance(a, torch.Tensor) and a.dtype == torch.int16 else a for a in targs]
        getattr(make_backend(pkg), base)(*targs)
    else:
        _lib.call(name, *work)
    torch.cuda.synchronize()
    return [None if w is None else w.cpu().numpy() for k, w in zip(sig, work) if k == "p"]


def rel_l2(a: np.ndarray, b: np.ndarray) -> float:
    a = a.astype(np.float64).ravel(); b = b.astype(np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def bits_equal(a: np.ndarray, b: np.ndarray) -> bool:
    """bitwise equality for float arrays (distinguishes -0/+0, matches NaN payloads)."""
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def half_ulp_distance(a16: np.ndarray, b16: np.ndarray) -> np.ndarray:
    """distance of two int16-viewed fp16 arrays in units of fp16 ulp: both mapped to the integer line on which consecutive
    halves are consecutive integers (sign-magnitude -> offset), so the result counts representable values between them"""
    def line(v):
        u = v.view(np.uint16).astype(np.int64)
        return np.where(u & 0x8000, -(u & 0x7fff), u & 0x7fff)
    return np.abs(line(np.ascontiguousarray(a16)) - line(np.ascontiguousarray(b16)))
