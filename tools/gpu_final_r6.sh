#!/bin/bash
# round 6, final collection (through gpurun): everything tools/gpu_round.sh collects on the final tree (tests, smoke, bench, kernel stats,
# PMC passes -> summary.json), the sanitizer driver, and the split-precision kernel's counters.   tools/gpu_final_r6.sh <tag>
cd $GRAFT_REPO_ROOT
TAG=${1:-r06b}
bash tools/gpu_round.sh $TAG > gpurun_out/${TAG}_round.log 2>&1
tail -30 gpurun_out/${TAG}_round.log
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/$TAG/pytest_gpu_all.log 2>&1; echo "pytest (no -x) rc=$?" >> gpurun_out/$TAG/pytest_gpu_all.log; tail -5 gpurun_out/$TAG/pytest_gpu_all.log
bash tools/gpu_asan.sh ${TAG}_asan > gpurun_out/${TAG}_asan.log 2>&1; tail -30 gpurun_out/${TAG}_asan.log
bash tools/gpu_split2.sh ${TAG}_split > gpurun_out/${TAG}_split.log 2>&1; tail -30 gpurun_out/${TAG}_split.log
