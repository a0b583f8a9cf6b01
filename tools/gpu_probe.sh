#!/bin/bash
# tools/gpu_probe.sh <tag> <probe> [probe...] [-- pytest args]: run prebuilt microbenchmarks of tools/probe on the GPU box (through gpurun)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
while [ $# -gt 0 ] && [ "$1" != "--" ]; do
  p=$1; shift
  timeout 600 tools/probe/$p > $OUT/$p.txt 2>&1; echo "rc=$?" >> $OUT/$p.txt
  cat $OUT/$p.txt
done
if [ "$1" == "--" ]; then shift; timeout 1500 python -m pytest "$@" -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log; fi
