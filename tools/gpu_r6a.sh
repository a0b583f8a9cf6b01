#!/bin/bash
# round 6, first GPU pass: full GPU tests, the exact / contracted reference sweeps, smoke, the bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06a
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q -x --deselect tests/test_refhip_gpu.py::test_operator_has_the_bits_of_the_references_kernel_built_without_contraction > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
timeout 900 python tools/refhip_sweep.py --exact > $OUT/refhip_sweep_exact.txt 2>&1
timeout 900 python tools/refhip_sweep.py > $OUT/refhip_sweep.txt 2>&1
timeout 900 python -m pytest tests/test_refhip_gpu.py -q -k without_contraction > $OUT/pytest_exact.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_exact.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/bench.err
tail -5 $OUT/pytest_gpu.log; tail -3 $OUT/refhip_sweep_exact.txt; tail -3 $OUT/pytest_exact.log; tail -2 $OUT/smoke.log; tail -3 $OUT/bench.err; head -c 1500 $OUT/bench.json
