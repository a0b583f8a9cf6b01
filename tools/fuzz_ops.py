"""Every operator case group of tests/cases.py re-generated with other random seeds and run HIP against oracle with the
comparison of tests/test_ops_gpu.py (bit-exact / per-case tolerance).  Run on the GPU box:  python tools/fuzz_ops.py [first] [count]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from tests import cases, test_ops_gpu
from tests.util import run_op

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 5
bad = total = 0
for off in range(first, first + count):
    cases.SEED_OFFSET = 1000 * off
    for cid, op, args, tol in cases.all_cases():
        total += 1
        try:
            want = run_op("oracle", op, *args)
            got = run_op("hip", op, *args)
            if op == "march_rays_train":
                g, gc = test_ops_gpu._regroup_train(got); w, wc = test_ops_gpu._regroup_train(want)
                assert np.array_equal(gc, wc) and g.keys() == w.keys()
                for rid in w:
                    for a, b in zip(g[rid], w[rid]):
                        assert test_ops_gpu.bits_equal(a, b), f"ray {rid}"
            else:
                test_ops_gpu._compare(cid, op, got, want, tol)
        except AssertionError as e:
            bad += 1
            print(f"seed offset {cases.SEED_OFFSET} {cid} ({op}): {str(e)[:200]}")
print(f"{total} cases over {count} seed offsets, {bad} failures")
sys.exit(1 if bad else 0)
