#!/bin/bash
# tools/gpu_evidence.sh <tag>: the randomised differential runs and the kernel-coverage audit on the current tree (through gpurun)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python tools/fuzz_ops.py 20 16 > $OUT/fuzz_ops_long.txt 2>&1
timeout 900 python tools/geo/fuzz_frames.py 4000 7 > $OUT/fuzz_frames_long.txt 2>&1
timeout 600 python tools/fuzz_dropin.py 100 60 > $OUT/fuzz_dropin.txt 2>&1
timeout 600 python tools/fuzz_indirect.py > $OUT/fuzz_indirect.txt 2>&1
timeout 600 python tools/fuzz_train_vs_infer.py > $OUT/fuzz_train_vs_infer.txt 2>&1
timeout 600 python tests/tools/fuzz_shade.py > $OUT/fuzz_shade.txt 2>&1
timeout 600 python tools/geo/fuzz_sequence.py > $OUT/fuzz_sequence.txt 2>&1
timeout 1500 tools/kernel_coverage.sh > $OUT/kernel_coverage.txt 2>&1
cp gpurun_out/coverage/never_launched.txt $OUT/ 2>/dev/null
for f in $OUT/fuzz_*.txt; do echo "== $f"; tail -2 $f; done
tail -25 $OUT/kernel_coverage.txt
