/*
 * TEST INFRASTRUCTURE -- never linked into, imported by, or shipped with the product path.
 *
 * The DEVICE counterpart of kernel_keywords.h: the same slices of the reference's kernels (/root/reference/<ext>/src/<ext>.cu, read
 * where they lie, host launchers dropped) compiled by hipcc for gfx950 and run ON THE GPU.  HIP's kernel language is a superset of
 * what the kernels use, so nothing of theirs is restated here: `threadIdx` / `blockIdx` / `blockDim`, `atomicAdd` on float / int,
 * `__expf` / `__sinf`, `__half` / `__half2` are hipcc's own, `at::Half` is c10's own header (with its device conversions to `__half`),
 * FMA contraction is the compiler's default for device code -- the one deviation from the op-by-op CPU build that a real CUDA build
 * has as well.  Three things are ours:
 *   - the kernels are entered through ONE generic __global__ wrapper that calls them as device functions (`__global__` is re-defined
 *     to `__device__` for the slices): ref_entry.cpp's launch lambdas then serve both builds;
 *   - `atomicAdd(__half2*, __half2)` / `atomicAdd(__half*, __half)`: CUDA overloads HIP spells `unsafeAtomicAdd`; forwarded;
 *   - grid shapes come from ref_entry.cpp (the reference's host functions).
 * This is still NOT a build of the reference's extension (no nvcc, no CUDA headers here); it is the reference's kernel text compiled by
 * the GPU's own compiler, which is as close as this image gets.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <type_traits>

#include <c10/util/Half.h>
namespace at { using c10::Half; }

template <typename F>
__global__ void emu_kernel(F f) { f(); }

template <typename F>
static inline void emu_launch(uint32_t grid_x, uint32_t grid_y, uint32_t block_x, F body) {
    if (grid_x == 0 || grid_y == 0) return;
    hipLaunchKernelGGL(emu_kernel<F>, dim3(grid_x, grid_y), dim3(block_x), 0, 0, body);
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) fprintf(stderr, "[oracle/ref device build] launch failed: %s\n", hipGetErrorString(e));
}
static inline uint32_t emu_blocks(uint32_t n, uint32_t block) { return (n + block - 1) / block; }
#define EMU_LAMBDA [=] __device__

/* CUDA's half atomics under the names the kernels call */
static inline __device__ __half2 atomicAdd(__half2* a, __half2 v) { return unsafeAtomicAdd(a, v); }
static inline __device__ __half atomicAdd(__half* a, __half v) { return unsafeAtomicAdd(a, v); }

/* from here on the slices' `__global__` functions are device functions called by emu_kernel */
#undef __global__
#define __global__ __device__
