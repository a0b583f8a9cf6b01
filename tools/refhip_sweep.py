"""TEST TOOLING: every operator case of tests/cases.py through (a) libenvidr_amd.so and (b) the REFERENCE's own kernels compiled by hipcc
for this GPU (oracle/_ref/libenvidr_ref_hip.so, oracle/ref/device_keywords.h), on the same device arrays: what is identical, what moves by
how much.  `--exact` compares with libenvidr_ref_hip_exact.so instead (the same kernel text built with -ffp-contract=off, the way the
product is built).  Run through gpurun:  python tools/refhip_sweep.py [--exact]"""
import re
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import clib  # noqa: E402
from tests import cases  # noqa: E402
from tests.util import bits_equal, run_op  # noqa: E402


def f64(a):
    return a.view(np.float16).astype(np.float64) if a.dtype == np.int16 else a.astype(np.float64)


def pointer_names():
    """operator -> names of its pointer parameters in declaration order (include/envidr_amd.h)"""
    text = (Path(__file__).resolve().parents[1] / "include" / "envidr_amd.h").read_text()
    out = {}
    for m in re.finditer(r"int\s+envidr_(\w+)\s*\(([^;]*?)\)\s*;", text, re.S):
        out[m.group(1)] = [re.split(r"[\s\*]+", p.strip())[-1] for p in m.group(2).split(",") if "*" in p]
    return out


def main():
    exact = "--exact" in sys.argv
    lib = clib.ref_hip_exact() if exact else clib.ref_hip()
    backend = "refhip_exact" if exact else "refhip"
    names = pointer_names()
    ident = total = 0
    for cid, op, args, tol in cases.all_cases():
        if not lib.has(op):
            continue
        ours, theirs = run_op("hip", op, *args), run_op(backend, op, *args)
        notes = []
        for k, (a, b) in enumerate(zip(ours, theirs)):
            if a is None or bits_equal(a, b):
                continue
            nm = names.get(op, [])
            nm = nm[k] if k < len(nm) else f"arg{k}"
            if a.dtype.kind in "iu" and a.dtype != np.int16:
                notes.append(f"{nm} INTEGER {int((a != b).sum())}/{a.size}")
            else:
                d = np.abs(f64(a) - f64(b))
                notes.append(f"{nm} {a.dtype} n={int((d > 0).sum())}/{a.size} max={d.max():.2e} rel={np.linalg.norm(d) / max(np.linalg.norm(f64(b)), 1e-30):.1e}")
        total += 1
        ident += not notes
        print(f"{cid:40s} {op:34s} " + ("IDENTICAL" if not notes else "; ".join(notes)))
    print(f"{ident} of {total} cases bit-identical between libenvidr_amd.so and the reference's kernels compiled by hipcc"
          + (" with -ffp-contract=off" if exact else " (default contraction)"))


if __name__ == "__main__":
    main()
