"""TEST INFRASTRUCTURE: ctypes access to
    * oracle/_build/libenvidr_oracle.so -- our plain-C restatement (oracle/c/envidr_oracle.c), prefix `oracle_`
    * oracle/_ref/libenvidr_ref.so      -- the reference's own kernel bodies on the CPU (oracle/ref), prefix `ref_`
    * oracle/_ref/libenvidr_ref_hip.so  -- the same kernel text compiled by hipcc, run on the GPU (device pointers)
Both expose the C-ABI of include/envidr_amd.h minus the stream argument, on host pointers, so the
same argument tuples drive the oracle, the reference bodies and (via envidr_amd._lib) the HIP library.
"""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

from envidr_amd._lib import SIGNATURES, argtypes  # the signature table only; no GPU needed

HERE = Path(__file__).resolve().parent
ORACLE_LIB = HERE / "_build" / "libenvidr_oracle.so"
REF_LIB = HERE / "_ref" / "libenvidr_ref.so"


def build_oracle(verbose: bool = False) -> Path:
    src = HERE / "c" / "envidr_oracle.c"
    if not ORACLE_LIB.exists() or ORACLE_LIB.stat().st_mtime < src.stat().st_mtime:
        r = subprocess.run(["make", "-C", str(HERE / "c")], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
        if verbose:
            print(f"[oracle] built {ORACLE_LIB}")
    return ORACLE_LIB


class HostLib:
    """Calls `<prefix><name>(...)` with numpy arrays (passed by address), numbers and None."""

    def __init__(self, path: Path, prefix: str):
        self.path, self.prefix = path, prefix
        self.lib = ctypes.CDLL(str(path))
        for name, sig in SIGNATURES.items():
            fn = getattr(self.lib, prefix + name, None)
            if fn is not None:
                fn.argtypes = argtypes(sig, with_stream=False)
                fn.restype = ctypes.c_int

    def has(self, name: str) -> bool:
        return hasattr(self.lib, self.prefix + name)

    def call(self, name: str, *args) -> None:
        sig = SIGNATURES[name]
        assert len(args) == len(sig), f"{name}: expected {len(sig)} args, got {len(args)}"
        conv = []
        for kind, a in zip(sig, args):
            if kind == "p":
                if a is None:
                    conv.append(None)
                elif isinstance(a, int):
                    conv.append(a)                                   # a device address (ref_hip)
                else:
                    assert isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"], f"{name}: need contiguous ndarray"
                    conv.append(a.ctypes.data)
            elif kind == "f":
                conv.append(float(a))
            else:
                conv.append(int(a))
        rc = getattr(self.lib, self.prefix + name)(*conv)
        if rc != 0:
            raise RuntimeError(f"{self.prefix}{name} returned {rc}")


_oracle: HostLib | None = None
_ref: HostLib | None = None


def oracle() -> HostLib:
    global _oracle
    if _oracle is None:
        _oracle = HostLib(build_oracle(), "oracle_")
    return _oracle


REF_HIP_LIB = HERE / "_ref" / "libenvidr_ref_hip.so"
_ref_hip: HostLib | None = None


def ref_hip_available() -> bool:
    return REF_HIP_LIB.exists()


def ref_hip() -> HostLib:
    """The reference's kernels compiled by hipcc for the GPU (oracle/ref/device_keywords.h).  Same entry points as ref(), but the
    pointer arguments are DEVICE addresses: call it through tests/util.py run_op("refhip", ...), which stages the arrays."""
    global _ref_hip
    if _ref_hip is None:
        if not REF_HIP_LIB.exists():
            raise FileNotFoundError(f"{REF_HIP_LIB} not built; run `python oracle/ref/build_ref.py` where /root/reference exists")
        _ref_hip = HostLib(REF_HIP_LIB, "ref_")
    return _ref_hip


REF_HIP_EXACT_LIB = HERE / "_ref" / "libenvidr_ref_hip_exact.so"
_ref_hip_exact: HostLib | None = None


def ref_hip_exact_available() -> bool:
    return REF_HIP_EXACT_LIB.exists()


def ref_hip_exact() -> HostLib:
    """ref_hip() built with `-ffp-contract=off`: the reference's kernel text evaluated operation by operation on the GPU, the way
    libenvidr_amd.so is built -- the bit-for-bit comparison partner (tests/test_refhip_gpu.py)."""
    global _ref_hip_exact
    if _ref_hip_exact is None:
        if not REF_HIP_EXACT_LIB.exists():
            raise FileNotFoundError(f"{REF_HIP_EXACT_LIB} not built; run `python oracle/ref/build_ref.py` where /root/reference exists")
        _ref_hip_exact = HostLib(REF_HIP_EXACT_LIB, "ref_")
    return _ref_hip_exact


def ref_available() -> bool:
    return REF_LIB.exists()


def ref() -> HostLib:
    """The reference's kernel bodies (prebuilt in the container that has /root/reference)."""
    global _ref
    if _ref is None:
        if not REF_LIB.exists():
            raise FileNotFoundError(f"{REF_LIB} not built; run `python oracle/ref/build_ref.py` where /root/reference exists")
        _ref = HostLib(REF_LIB, "ref_")
    return _ref
