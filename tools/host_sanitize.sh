#!/bin/bash
# The CPU test suite against the host-sanitized build of the library (tools/build_host_asan.py): AddressSanitizer + UBSan on the host half
# of every translation unit -- the weight packers, descriptor validation, scratch bookkeeping and error paths that ctypes reaches without
# a GPU.  Usage: bash tools/host_sanitize.sh [out.txt]
set -u
cd "$(dirname "$0")/.."
OUT=${1:-/tmp/host_sanitize.txt}
python tools/build_host_asan.py > /tmp/host_asan_build.log 2>&1 || { tail -20 /tmp/host_asan_build.log; exit 1; }
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so)
rm -f /tmp/hostasan.* /tmp/hostubsan.*
ENVIDR_AMD_LIB=$PWD/tools/asan_host/libenvidr_amd_hostasan.so LD_PRELOAD=$RT \
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:log_path=/tmp/hostasan UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:log_path=/tmp/hostubsan \
  python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -3 > "$OUT"
echo "sanitizer report files: $(ls /tmp/hostasan.* /tmp/hostubsan.* 2>/dev/null | wc -l)" >> "$OUT"
cat "$OUT"
