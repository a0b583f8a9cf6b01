/*
 * TEST INFRASTRUCTURE -- never linked into, imported by, or shipped with the product path.
 *
 * Keyword/intrinsic definitions that let the *bodies* of the reference's device kernels
 * (/root/reference/<ext>/src/<ext>.cu, read where they lie, never copied into this repo) be
 * compiled by g++ as ordinary host C++ and driven one emulated thread at a time.
 *
 * This is NOT a build of the reference's CUDA extension: that needs nvcc, the CUDA runtime
 * headers and a CUDA device, none of which exist in this image (DESIGN.md "Oracle pinning"
 * says so plainly).  What it gives us is the reference authors' own arithmetic -- the same
 * expressions, in the same order, with the same float/double/int mixing -- executed on the
 * CPU, which is what SURVEY.md section 8(c) prescribes as the level-0 oracle.
 *
 * Known, documented deviations from a real CUDA run:
 *   - __expf/__sinf map to libm expf/sinf (device fast-math differs by a few ulp anyway);
 *   - no FMA contraction (g++ -ffp-contract=off); nvcc would contract a*b+c;
 *   - atomics are serial, therefore deterministic.
 */
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <type_traits>

#define __global__
#define __device__
#define __host__
#define __restrict__

struct EmuDim3 { uint32_t x = 0, y = 0, z = 0; };
static thread_local EmuDim3 threadIdx, blockIdx, blockDim;

using std::max;
using std::min;

/* serial stand-ins for the device atomics the kernels use */
template <typename T> static inline T atomicAdd(T* addr, T v) { T old = *addr; *addr = old + v; return old; }
static inline uint32_t atomicAdd(int* addr, uint32_t v) { int old = *addr; *addr = old + (int)v; return (uint32_t)old; }

/* at::Half IS c10::Half: the real header from the torch wheel of this image (torch/include/c10/util/Half.h -> torch/headeronly/util/Half.h,
 * Half-inl.h), found through torch.utils.cpp_extension.include_paths() by build_ref.py.  The `scalar_t = at::Half` instantiations of the
 * reference's kernel templates (AT_DISPATCH_FLOATING_TYPES_AND_HALF, hashencoder.cu:747,778 / gridencoder.cu:443,474) therefore narrow
 * exactly where a real build narrows -- conversions, `Half op Half`, `Half op float`, `Half += float` are c10's own definitions, not ours.
 * (Round 4 carried a 60-line restatement of the class here; it is gone.) */
#include <c10/util/Half.h>
namespace at { using c10::Half; }
/* CUDA's __half has no host definition without the CUDA headers: a 16-bit carrier that converts to and from c10::Half / float, which is all
 * the kernels ask of it (`(__half)(w * g)`, a pair of them for the packed atomic, hashencoder.cu:24-26 handing an at::Half to the scalar
 * atomic).  The arithmetic of the atomics below goes through c10::Half's own conversions. */
struct __half {
    c10::Half h;
    __half() = default;
    __half(float f) : h(f) {}
    __half(c10::Half v) : h(v) {}
    operator c10::Half() const { return h; }
    explicit operator float() const { return (float)h; }
};
struct __half2 { __half x, y; };
/* serial stand-ins for the half atomics: fp16 addition of each component, round to nearest even */
static inline __half atomicAdd(__half* a, __half v) { const __half old = *a; *a = __half((float)old.h + (float)v.h); return old; }
static inline void atomicAdd(__half2* a, __half2 v) { atomicAdd(&a->x, v.x); atomicAdd(&a->y, v.y); }

#define __sinf sinf
#define __expf expf

/* Run `body()` once per emulated thread of a 1-D or (x, y) grid. */
template <typename F>
static inline void emu_launch(uint32_t grid_x, uint32_t grid_y, uint32_t block_x, F body) {
    blockDim.x = block_x; blockDim.y = 1; blockDim.z = 1;
    for (uint32_t by = 0; by < grid_y; ++by)
        for (uint32_t bx = 0; bx < grid_x; ++bx)
            for (uint32_t tx = 0; tx < block_x; ++tx) {
                blockIdx.x = bx; blockIdx.y = by; threadIdx.x = tx;
                body();
            }
}
static inline uint32_t emu_blocks(uint32_t n, uint32_t block) { return (n + block - 1) / block; }
#define EMU_LAMBDA [&]          /* (the device build captures by value: device_keywords.h) */
