#!/bin/bash
# AddressSanitizer pass over the library on the GPU (through gpurun), then the synchronisation stress loop on the product library.
#   tools/gpu_asan.sh <tag>        needs tools/asan/libenvidr_amd_asan.so (python tools/build_asan.py, in the container)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cp tools/asan/build.log $OUT/asan_build.log 2>/dev/null
# 1. the stress loop on the product library (no sanitizer: full speed, 1000 repetitions)
timeout 1500 python tools/stress_sync.py 1000 > $OUT/stress_sync.txt 2>&1; echo "stress rc=$?" >> $OUT/stress_sync.txt; tail -8 $OUT/stress_sync.txt
# 2. the sanitizer build: device code instrumented (xnack+), driven without PyTorch (tools/asan_driver.py says why)
if [ -f tools/asan/libenvidr_amd_asan.so ]; then
  RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
  ( export HSA_XNACK=1 LD_PRELOAD=$RT LD_LIBRARY_PATH=/opt/rocm/lib
    export ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:protect_shadow_gap=0:log_path=$OUT/asan_report
    timeout 3000 python tools/asan_driver.py > $OUT/asan_driver.txt 2>&1; echo "driver rc=$?" >> $OUT/asan_driver.txt )
  tail -25 $OUT/asan_driver.txt
  for f in $OUT/asan_report*; do [ -f "$f" ] && head -60 "$f"; done
  echo "sanitizer reports: $(ls $OUT/asan_report* 2>/dev/null | wc -l)" | tee $OUT/asan_summary.txt
else
  echo "tools/asan/libenvidr_amd_asan.so not built" | tee $OUT/asan_summary.txt
fi
