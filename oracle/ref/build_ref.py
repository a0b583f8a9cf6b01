"""TEST INFRASTRUCTURE -- builds oracle/_ref/libenvidr_ref.so, the reference's OWN kernel bodies
driven on the CPU (SURVEY.md 8c level-0 oracle).  Runs only where /root/reference exists (the
build container); the GPU box only ever sees the prebuilt .so that travels with the snapshot.

What it does
  1. reads each /root/reference/<ext>/src/<ext>.cu *where it lies*;
  2. drops the `#include` lines and every host-side function (the column-0 `void f(...) {...}`
     definitions, which contain `<<<...>>>` launches / AT_DISPATCH), keeping helpers + kernels;
  3. writes those slices to a TEMPORARY directory (never into the repo, never onto the GPU box);
  4. compiles oracle/ref/ref_entry.cpp (ours) with g++ against kernel_keywords.h (ours), with
     `-I <tmp>` so the `#include "<ext>_kernels.inc"` lines resolve; output -> oracle/_ref/.

This is not a build of the reference's CUDA extension (that is unbuildable here: no nvcc, no CUDA
headers, no CUDA device) -- see kernel_keywords.h for the exact list of deviations.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
OUT_DIR = HERE.parent / "_ref"
OUT_LIB = OUT_DIR / "libenvidr_ref.so"
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


def build(verbose: bool = True) -> Path | None:
    if not reference_available():
        if verbose:
            print(f"[oracle/ref] {REFERENCE} not present: keeping prebuilt {OUT_LIB.name} "
                  f"({'found' if OUT_LIB.exists() else 'absent'})")
        return OUT_LIB if OUT_LIB.exists() else None
    srcs = [REFERENCE / e / "src" / f"{e}.cu" for e in EXTENSIONS]
    deps = srcs + [HERE / "ref_entry.cpp", HERE / "kernel_keywords.h", Path(__file__)]
    if OUT_LIB.exists() and all(d.stat().st_mtime <= OUT_LIB.stat().st_mtime for d in deps):
        return OUT_LIB
    OUT_DIR.mkdir(parents=True, exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="envidr_ref_slices_") as tmp:
        for e, src in zip(EXTENSIONS, srcs):
            (Path(tmp) / f"{e}_kernels.inc").write_text(slice_kernels(src.read_text()))
        cmd = ["g++", "-O2", "-std=c++17", "-w", "-fPIC", "-shared", "-ffp-contract=off", "-fvisibility=hidden",
               "-I", tmp, "-I", str(HERE), str(HERE / "ref_entry.cpp"), "-o", str(OUT_LIB)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("reference kernel-body build failed:\n" + r.stderr[-6000:])
    if verbose:
        print(f"[oracle/ref] built {OUT_LIB} ({OUT_LIB.stat().st_size >> 10} KiB)")
    return OUT_LIB


if __name__ == "__main__":
    build()
