#!/bin/bash
# which environment lets PyTorch initialise the GPU with the AddressSanitizer runtime preloaded (tools/gpu_asan.sh uses the first that works)
cd $GRAFT_REPO_ROOT
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
TL=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
export HSA_XNACK=1 ENVIDR_AMD_LIB=$GRAFT_REPO_ROOT/tools/asan/libenvidr_amd_asan.so
SNIP='import torch; x = torch.zeros(1000, 3).cuda(); torch.cuda.synchronize(); from envidr_amd import _lib; import numpy as np; o = torch.empty(1000, device="cuda"); f = torch.empty(1000, device="cuda"); d = torch.ones(1000, 3, device="cuda"); aabb = torch.tensor([-1, -1, -1, 1, 1, 1.0], device="cuda"); _lib.call("near_far_from_aabb", x, d, aabb, 1000, 0.2, o, f); torch.cuda.synchronize(); print("WORKS", float(o.sum()))'
for combo in "A" "B" "C" "D"; do
  case $combo in
    A) export LD_LIBRARY_PATH=$TL; OPTS="detect_leaks=0:protect_shadow_gap=0";;
    B) export LD_LIBRARY_PATH=$TL; OPTS="detect_leaks=0:protect_shadow_gap=0:allocator_may_return_null=1:max_allocation_size_mb=8192";;
    C) export LD_LIBRARY_PATH=/opt/rocm/lib/asan:/opt/rocm/lib:$TL; OPTS="detect_leaks=0:protect_shadow_gap=0";;
    D) export LD_LIBRARY_PATH=/opt/rocm/lib/asan:/opt/rocm/lib:$TL; OPTS="detect_leaks=0:protect_shadow_gap=0:allocator_may_return_null=1";;
  esac
  echo "== combo $combo: LD_LIBRARY_PATH=$LD_LIBRARY_PATH ASAN_OPTIONS=$OPTS"
  LD_PRELOAD=$RT ASAN_OPTIONS=$OPTS timeout 300 python -c "$SNIP" 2>&1 | tail -6
done
