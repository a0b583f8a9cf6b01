#!/bin/bash
# shading time of the split frame under the ablation variants of the fused-pair kernel (tools/geo/build_variants.py s2_*): tools/gpu_s2_variants.sh <tag> <variant...>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
timeout 600 python -m pytest tests/test_split_gpu.py -x -q > $OUT/pytest_split.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_split.log; tail -4 $OUT/pytest_split.log
: > $OUT/variants.txt
for rep in 1 2; do
  echo "base: $(timeout 300 python tools/geo/split_probe.py 2>/dev/null | tail -1)" >> $OUT/variants.txt
  for v in "$@"; do
    echo "$v: $(ENVIDR_AMD_LIB=$GRAFT_REPO_ROOT/tools/geo/variants/$v.so timeout 300 python tools/geo/split_probe.py 2>/dev/null | tail -1)" >> $OUT/variants.txt
  done
done
echo "v1: $(timeout 300 python tools/geo/split_probe.py --v1 2>/dev/null | tail -1)" >> $OUT/variants.txt
cat $OUT/variants.txt
