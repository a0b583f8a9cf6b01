#!/bin/bash
# A/B of kernel variants (tools/geo/build_variants.py) on the cold and the hinted headline frame: tools/gpu_variants.sh <tag> name...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
for V in "$@"; do
  for MODE in "" "--hinted"; do
    ENVIDR_AMD_LIB=$GRAFT_REPO_ROOT/tools/geo/variants/$V.so timeout 300 python bench.py --headline-only $MODE --steps 10 --warmup 2 > $OUT/$V$MODE.json 2> $OUT/$V$MODE.err
    python - <<PY
import json
try:
    j = json.loads(open("$OUT/$V$MODE.json").read().strip().splitlines()[-1])
    print("$V", "$MODE" or "cold", "ms/frame %.3f geometry %.3f shading %.3f evaluated %d" % (j["ms_per_step"], j["frame"]["geometry_ms"], j["frame"]["shading_ms"], j["config"]["samples_evaluated_per_frame"]))
except Exception as e:
    print("$V", "$MODE", "failed", e, open("$OUT/$V$MODE.err").read()[-500:])
PY
  done
done
