"""Build recipe for the gfx950 shared library (`libenvidr_amd.so`) -- plain hipcc, no torch
extension machinery: the library's boundary is a C ABI (include/envidr_amd.h), loaded with
ctypes by envidr_amd._lib.  hipcc cross-compiles gfx950 code objects without a GPU present.

    python -m envidr_amd.build            # incremental
    python -m envidr_amd.build --force    # rebuild everything
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = CSRC / "build"
LIB = PKG / "libenvidr_amd.so"

ARCH = "gfx950"
# -ffp-contract=off: the marching / grid-index arithmetic must round exactly like the reference's
#   expressions evaluated op by op (DESIGN.md "bit-exact marching"); fused multiply-adds are
#   written explicitly (fmaf / MFMA) where they are wanted.
# -munsafe-fp-atomics: fp32 scatter-adds become one global_atomic_add_f32 instead of a CAS loop.
HIPCC_FLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++20", "-fPIC", "-ffp-contract=off",
    "-munsafe-fp-atomics", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas",
]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP kernels cannot be built")
    return exe


def sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def _compile(src: Path, force: bool, extra: list[str]) -> Path:
    obj = OBJ / (src.stem + ".o")
    deps = [src, *CSRC.glob("*.h"), *CSRC.glob("*.inc"), PKG.parent / "include" / "envidr_amd.h",
            PKG.parent / "include" / "envidr_render.h"]
    deps = [d for d in deps if d.exists()]
    if force or _stale(obj, deps):
        cmd = [hipcc(), *HIPCC_FLAGS, *extra, "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stderr[-8000:]}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = True, extra_flags: list[str] | None = None) -> Path:
    OBJ.mkdir(parents=True, exist_ok=True)
    extra = list(extra_flags or [])
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as pool:
        objs = list(pool.map(lambda s: _compile(s, force, extra), srcs))
    if force or _stale(LIB, objs):
        # no rpath on purpose: inside a torch process libamdhip64.so.7 is already mapped (torch
        # bundles it) and the loader reuses it by SONAME; standalone use falls back to /opt/rocm.
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fno-gpu-rdc",
               *map(str, objs), "-o", str(LIB)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-8000:]}")
    if verbose:
        print(f"[envidr_amd.build] {LIB} ({LIB.stat().st_size >> 10} KiB, {len(srcs)} sources)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, extra_flags=[a for a in sys.argv[1:] if a.startswith("-D")])
