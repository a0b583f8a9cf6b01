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

/* at::Half as c10 defines it (c10/util/Half.h, Half-inl.h), restated: 16 bits of storage, conversions to and from float
 * with round-to-nearest-even, and arithmetic that is ALWAYS done in float --
 *     Half op Half   -> float result converted back to Half      (operator+,-,*,/ and the compound forms)
 *     Half op float  -> float                                     (no narrowing)
 *     Half += float  -> resolves to operator+=(Half&, const Half&): the float is narrowed FIRST, then added, then narrowed
 *     comparisons    -> through the implicit conversion to float
 * so that the `scalar_t = at::Half` instantiations of the reference's kernel templates (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
 * hashencoder.cu:747,778 / gridencoder.cu:443,474) narrow exactly where the real build narrows.  __half is the same type here
 * (the kernels only construct it from float and pack two of them for the paired atomic). */
struct __half;
namespace at {
struct Half {
    uint16_t bits;
    static uint16_t from_float(float f) {
        uint32_t x; std::memcpy(&x, &f, 4);
        const uint32_t sign = (x >> 16) & 0x8000u;
        x &= 0x7fffffffu;
        if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));      /* inf / nan */
        if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                          /* rounds to inf */
        if (x < 0x38800000u) {                                                                            /* subnormal half or zero */
            if (x < 0x33000000u) return (uint16_t)sign;                                                  /* < 2^-25: zero (ties to even: 2^-25 -> 0) */
            const int e = (int)(x >> 23);
            const uint32_t m = (x & 0x7fffffu) | 0x800000u;
            const int shift = 126 - e;                        /* 14 .. 24: bits to drop to reach units of 2^-24 */
            uint32_t r = m >> shift;
            const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
            if (rem > half || (rem == half && (r & 1u))) ++r;
            return (uint16_t)(sign | r);
        }
        uint32_t r = ((x - 0x38000000u) >> 13);               /* rebias exponent, keep 10 mantissa bits */
        const uint32_t rem = x & 0x1fffu;
        if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;   /* may carry into the exponent: still correct */
        return (uint16_t)(sign | r);
    }
    static float to_float(uint16_t h) {
        const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
        uint32_t x;
        if (e == 0) {
            if (m == 0) x = sign;
            else { float f = (float)m * 5.9604644775390625e-08f; std::memcpy(&x, &f, 4); x |= sign; }   /* m * 2^-24, exact */
        } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
        else x = sign | ((e + 112u) << 23) | (m << 13);
        float f; std::memcpy(&f, &x, 4); return f;
    }
    Half() = default;
    Half(float f) : bits(from_float(f)) {}
    inline Half(const __half& h);
    operator float() const { return to_float(bits); }
};
inline Half operator+(const Half& a, const Half& b) { return Half((float)a + (float)b); }
inline Half operator-(const Half& a, const Half& b) { return Half((float)a - (float)b); }
inline Half operator*(const Half& a, const Half& b) { return Half((float)a * (float)b); }
inline Half operator/(const Half& a, const Half& b) { return Half((float)a / (float)b); }
inline Half& operator+=(Half& a, const Half& b) { a = a + b; return a; }
inline Half& operator-=(Half& a, const Half& b) { a = a - b; return a; }
inline Half& operator*=(Half& a, const Half& b) { a = a * b; return a; }
inline float operator+(Half a, float b) { return (float)a + b; }
inline float operator-(Half a, float b) { return (float)a - b; }
inline float operator*(Half a, float b) { return (float)a * b; }
inline float operator/(Half a, float b) { return (float)a / b; }
inline float operator+(float a, Half b) { return a + (float)b; }
inline float operator-(float a, Half b) { return a - (float)b; }
inline float operator*(float a, Half b) { return a * (float)b; }
inline float operator/(float a, Half b) { return a / (float)b; }
inline float& operator+=(float& a, const Half& b) { return a += (float)b; }
}  // namespace at
/* CUDA's __half: a distinct 16-bit type that c10::Half converts to and from implicitly; the kernels only construct it from a
 * float (`(__half)(w * g)`), pack two for the paired atomic, and (hashencoder's helper) hand an at::Half to the scalar atomic */
struct __half {
    uint16_t bits;
    __half() = default;
    __half(float f) : bits(at::Half::from_float(f)) {}
    __half(at::Half h) : bits(h.bits) {}
    operator float() const { return at::Half::to_float(bits); }
};
inline at::Half::Half(const __half& h) : bits(h.bits) {}
struct __half2 { __half x, y; };
/* serial stand-ins for the half atomics: fp16 addition of each component, round to nearest even */
static inline __half atomicAdd(__half* a, __half v) { const __half old = *a; *a = __half((float)old + (float)v); return old; }
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
