"""TEST INFRASTRUCTURE -- builds oracle/_ref/libenvidr_ref.so, the reference's OWN kernel bodies
driven on the CPU (SURVEY.md 8c level-0 oracle).  Runs only where /root/reference exists (the
build container); the GPU box only ever sees the prebuilt .so that travels with the snapshot.

What it does
  1. reads each /root/reference/<ext>/src/<ext>.cu *where it lies*;
  2. drops the `#include` lines and every host-side function (the column-0 `void f(...) {...}`
     definitions, which contain `<<<...>>>` launches / AT_DISPATCH), keeping helpers + kernels;
  3. writes those slices to a TEMPORARY directory (never into the repo, never onto the GPU box);
  4. compiles oracle/ref/ref_entry.cpp (ours) with g++ against kernel_keywords.h (ours: CUDA keywords only -- at::Half is the
     REAL c10/util/Half.h of this image's torch wheel, include path from torch.utils.cpp_extension), with
     `-I <tmp>` so the `#include "<ext>_kernels.inc"` lines resolve; output -> oracle/_ref/;
  5. compiles the same translation unit a second time with `-ffp-contract=fast -mfma` -> libenvidr_ref_fma.so (contraction sweep);
  6. compiles it a third time with hipcc for gfx950 against device_keywords.h -> libenvidr_ref_hip.so: the kernels run on the GPU;
  7. and a fourth time, the same hipcc command plus `-ffp-contract=off` -> libenvidr_ref_hip_exact.so (bit-for-bit comparisons on the GPU).

This is not a build of the reference's CUDA extension (that is unbuildable here: no nvcc, no CUDA
headers, no CUDA device) -- see kernel_keywords.h for the exact list of deviations.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
OUT_DIR = HERE.parent / "_ref"
OUT_LIB = OUT_DIR / "libenvidr_ref.so"
# the same slices compiled the way nvcc compiles device code by default: a*b+c contracted into fused multiply-adds
# (`-ffp-contract=fast -mfma`).  Only tests/test_oracle_pinning.py's contraction sweep loads it: it shows which outputs of the
# marchers / grid encoders are invariant under the one rounding deviation a real CUDA build certainly has.
OUT_LIB_FMA = OUT_DIR / "libenvidr_ref_fma.so"
# the same slices compiled by hipcc FOR THE GPU (device_keywords.h): the reference's kernel text run on the MI355X itself -- device
# intrinsics, device atomics, the compiler's own FMA contraction.  Entry points take DEVICE pointers.  tests/test_refhip_gpu.py.
OUT_LIB_HIP = OUT_DIR / "libenvidr_ref_hip.so"
# ... and once more with `-ffp-contract=off`: the reference's expressions evaluated operation by operation ON THE GPU, which is how the
# product is built.  Against this one tests/test_refhip_gpu.py asserts BIT EQUALITY (marchers, compositors, grid / hash gathers and input
# gradients, frequency encoder); what is left to bounds there is what cannot be identical: atomic summation order, device transcendentals.
OUT_LIB_HIP_EXACT = OUT_DIR / "libenvidr_ref_hip_exact.so"
REFERENCE = Path(os.environ.get("ENVIDR_REFERENCE", "/root/reference"))
EXTENSIONS = ["raymarching", "hashencoder", "gridencoder", "freqencoder", "shencoder"]


def reference_available() -> bool:
    return all((REFERENCE / e / "src" / f"{e}.cu").is_file() for e in EXTENSIONS)


def _block_end(lines: list[str], start: int) -> int:
    """index of the line holding the brace that closes the first '{' found at/after `start`."""
    depth, seen = 0, False
    for i in range(start, len(lines)):
        for ch in lines[i]:
            if ch == "{":
                depth += 1
                seen = True
            elif ch == "}":
                depth -= 1
                if seen and depth == 0:
                    return i
    raise ValueError("unbalanced braces while slicing")


def slice_kernels(text: str) -> str:
    lines = text.splitlines()
    keep: list[str] = []
    i = 0
    while i < len(lines):
        ln = lines[i]
        if ln.startswith("#include"):
            i += 1
            continue
        if ln.startswith("void "):  # host launcher: drop it and a directly preceding template line
            if keep and keep[-1].startswith("template"):
                keep.pop()
            i = _block_end(lines, i) + 1
            continue
        keep.append(ln)
        i += 1
    body = "\n".join(keep)
    assert "<<<" not in body, "a host launcher survived slicing"
    return body + "\n"


def torch_include_dirs() -> list[str]:
    """where c10/util/Half.h lives (header-only; nothing of torch is linked)."""
    from torch.utils.cpp_extension import include_paths
    dirs = [d for d in include_paths() if (Path(d) / "c10" / "util" / "Half.h").is_file()]
    if not dirs:
        raise RuntimeError("c10/util/Half.h not found under torch's include paths")
    return dirs


def build(verbose: bool = True) -> Path | None:
    if not reference_available():
        if verbose:
            print(f"[oracle/ref] {REFERENCE} not present: keeping prebuilt {OUT_LIB.name} "
                  f"({'found' if OUT_LIB.exists() else 'absent'})")
        return OUT_LIB if OUT_LIB.exists() else None
    srcs = [REFERENCE / e / "src" / f"{e}.cu" for e in EXTENSIONS]
    deps = srcs + [HERE / "ref_entry.cpp", HERE / "kernel_keywords.h", HERE / "device_keywords.h", Path(__file__)]
    if all(o.exists() and all(d.stat().st_mtime <= o.stat().st_mtime for d in deps) for o in (OUT_LIB, OUT_LIB_FMA, OUT_LIB_HIP, OUT_LIB_HIP_EXACT)):
        return OUT_LIB
    OUT_DIR.mkdir(parents=True, exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="envidr_ref_slices_") as tmp:
        for e, src in zip(EXTENSIONS, srcs):
            (Path(tmp) / f"{e}_kernels.inc").write_text(slice_kernels(src.read_text()))
        inc = [x for d in torch_include_dirs() for x in ("-isystem", d)]
        for out, fp in ((OUT_LIB, ["-ffp-contract=off"]), (OUT_LIB_FMA, ["-ffp-contract=fast", "-mfma"])):
            cmd = ["g++", "-O2", "-std=c++17", "-w", "-fPIC", "-shared", *fp, "-fvisibility=hidden",
                   "-I", tmp, "-I", str(HERE), *inc, str(HERE / "ref_entry.cpp"), "-o", str(out)]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("reference kernel-body build failed:\n" + r.stderr[-6000:])
            if verbose:
                print(f"[oracle/ref] built {out} ({out.stat().st_size >> 10} KiB)")
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        for out, fp in ((OUT_LIB_HIP, []), (OUT_LIB_HIP_EXACT, ["-ffp-contract=off"])):
            cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-w", "-fPIC", "-shared", "-fno-gpu-rdc", "-x", "hip", *fp,
                   '-DENVIDR_REF_KEYWORDS="device_keywords.h"', "-I", tmp, "-I", str(HERE), *inc, str(HERE / "ref_entry.cpp"), "-o", str(out)]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("reference kernel device build failed:\n" + r.stderr[-6000:])
            if verbose:
                print(f"[oracle/ref] built {out} ({out.stat().st_size >> 10} KiB)")
    return OUT_LIB


if __name__ == "__main__":
    build()
