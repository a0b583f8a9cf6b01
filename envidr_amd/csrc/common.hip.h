// Shared helpers for the gfx950 kernels and their C-ABI launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/envidr_amd.h"

namespace envidr {

// ---- error plumbing ---------------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define ENVIDR_REQUIRE(cond, ...)                      \
    do {                                               \
        if (!(cond)) {                                 \
            ::envidr::set_error(__VA_ARGS__);          \
            return ENVIDR_EINVAL;                      \
        }                                              \
    } while (0)

static inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return ENVIDR_ELAUNCH;
    }
    return ENVIDR_OK;
}

static inline hipStream_t as_stream(envidr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// One wave64 per 64 work items; 256-thread workgroups = 4 waves = one wave per SIMD of a CU.
constexpr uint32_t kBlock = 256;

// ---- device helpers ---------------------------------------------------------------------------
// All marching / indexing arithmetic is compiled with FP contraction OFF so that every float op
// rounds exactly like the host-compiled reference expressions (DESIGN.md "bit-exact marching").

__host__ __device__ __forceinline__ uint32_t spread3(uint32_t v) {
    // 10 low bits of v -> every third bit (0, 3, 6, ...)
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__host__ __device__ __forceinline__ uint32_t morton_encode(uint32_t x, uint32_t y, uint32_t z) {
    return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
}
__host__ __device__ __forceinline__ uint32_t compact3(uint32_t v) {
    v &= 0x49249249u;
    v = (v | (v >> 2)) & 0xC30C30C3u;
    v = (v | (v >> 4)) & 0x0F00F00Fu;
    v = (v | (v >> 8)) & 0xFF0000FFu;
    v = (v | (v >> 16)) & 0x0000FFFFu;
    return v;
}

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

}  // namespace envidr
