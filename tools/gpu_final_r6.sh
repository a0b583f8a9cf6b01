#!/bin/bash
# round 6, final collection (through gpurun), most important first in case the call is cut short:
#   1. tools/gpu_round.sh core: GPU tests, smoke, bench, kernel stats, PMC passes -> summary.json
#   2. the sanitizer driver + stress loop, the split-precision kernel's counters
#   3. the whole GPU suite again without -x if the first run stopped at a failure (to see everything that fails)
#   4. tools/gpu_round.sh extras: probes, operator / training benches, randomised runs
# tools/gpu_final_r6.sh <tag>
cd $GRAFT_REPO_ROOT
TAG=${1:-r06b}
mkdir -p gpurun_out/$TAG
bash tools/gpu_round.sh $TAG full core > gpurun_out/${TAG}_round.log 2>&1
tail -30 gpurun_out/${TAG}_round.log
bash tools/gpu_asan.sh ${TAG}_asan > gpurun_out/${TAG}_asan.log 2>&1; tail -40 gpurun_out/${TAG}_asan.log
bash tools/gpu_split2.sh ${TAG}_split > gpurun_out/${TAG}_split.log 2>&1; tail -30 gpurun_out/${TAG}_split.log
if ! grep -q "pytest rc=0" gpurun_out/$TAG/pytest_gpu.log; then
  timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/$TAG/pytest_gpu_all.log 2>&1; echo "pytest (no -x) rc=$?" >> gpurun_out/$TAG/pytest_gpu_all.log; tail -15 gpurun_out/$TAG/pytest_gpu_all.log
fi
bash tools/gpu_round.sh $TAG quick extras > gpurun_out/${TAG}_extras.log 2>&1
tail -5 gpurun_out/${TAG}_extras.log
