"""TEST TOOLING: every operator case of tests/cases.py through (a) libenvidr_amd.so and (b) the REFERENCE's own kernels compiled by hipcc
for this GPU (oracle/_ref/libenvidr_ref_hip.so, oracle/ref/device_keywords.h), on the same device arrays: what is identical, what moves by
how much.  Run through gpurun:  python tools/refhip_sweep.py"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import clib  # noqa: E402
from tests import cases  # noqa: E402
from tests.util import bits_equal, run_op  # noqa: E402


def f64(a):
    return a.view(np.float16).astype(np.float64) if a.dtype == np.int16 else a.astype(np.float64)


def main():
    lib = clib.ref_hip()
    ident = total = 0
    for cid, op, args, tol in cases.all_cases():
        if not lib.has(op):
            continue
        ours, theirs = run_op("hip", op, *args), run_op("refhip", op, *args)
        notes = []
        for k, (a, b) in enumerate(zip(ours, theirs)):
            if a is None or bits_equal(a, b):
                continue
            if a.dtype.kind in "iu" and a.dtype != np.int16:
                notes.append(f"arg{k} INTEGER {int((a != b).sum())}/{a.size}")
            else:
                d = np.abs(f64(a) - f64(b))
                notes.append(f"arg{k} {a.dtype} n={int((d > 0).sum())}/{a.size} max={d.max():.2e} rel={np.linalg.norm(d) / max(np.linalg.norm(f64(b)), 1e-30):.1e}")
        total += 1
        ident += not notes
        print(f"{cid:40s} {op:34s} " + ("IDENTICAL" if not notes else "; ".join(notes)))
    print(f"{ident} of {total} cases bit-identical between libenvidr_amd.so and the reference's kernels compiled by hipcc")


if __name__ == "__main__":
    main()
