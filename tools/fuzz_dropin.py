"""tests/test_dropin_gpu.py::test_fused_vs_operator_loop_on_random_scenes (random blobby scenes, random rays incl. origins inside the box,
random knobs: the fused pipeline against the reference-shaped operator loop) over many more seeds.  Run on the GPU box:
    python tools/fuzz_dropin.py [first] [count]"""
import sys, traceback
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tests.test_dropin_gpu import test_fused_vs_operator_loop_on_random_scenes as one

first = int(sys.argv[1]) if len(sys.argv) > 1 else 4
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for seed in range(first, first + count):
    try:
        one(seed)
    except AssertionError as e:
        bad += 1
        print(f"seed {seed}: {str(e)[:200]}")
    except Exception as e:     # noqa: BLE001
        bad += 1
        print(f"seed {seed}: {type(e).__name__} {str(e)[:200]}")
print(f"{count} seeds, {bad} with findings")
sys.exit(1 if bad else 0)
