#!/bin/bash
# tools/geo/run_probe.sh <tag> [variant...]: time every built variant of the geometry kernel on the GPU box
cd $GRAFT_REPO_ROOT
TAG=${1:-geo}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
VARS=${@:-$(ls tools/geo/variants/*.so | xargs -n1 basename | sed 's/\.so//')}
for v in $VARS; do
  ENVIDR_AMD_LIB=$GRAFT_REPO_ROOT/tools/geo/variants/$v.so timeout 300 python tools/geo/eval_probe.py 800 2>&1 | tail -2 | tee -a $OUT/probe.log
done
