// Row-major [B, C] fp32 arrays <-> one row per lane, with the GLOBAL side coalesced.
//
// The streaming encoders (frequency, spherical harmonics, IDE) compute a whole row of C outputs per lane.  Stored straight
// from the lanes, every store instruction is 64 four-byte writes C * 4 bytes apart: each touches 64 cache lines and moves
// 4 useful bytes on each.  wave_store_rows() passes the wave's 64 x C tile through LDS (odd row pitch: conflict-free column
// writes) and writes the tile -- one contiguous run of 64 C floats in memory -- as 16 bytes per lane, lanes consecutive.
// A wave's LDS instructions execute in order, so within the wave-private slab no barrier is needed; the wave_barrier() only
// keeps the compiler from moving the reads above the writes (it sees no data dependence between different lanes).
#pragma once
#include "common.hip.h"

namespace envidr {

template <int C> constexpr int row_pitch() { return (C % 2 == 0) ? C + 1 : C; }
template <int C> constexpr int wave_tile_floats() { return 64 * row_pitch<C>(); }

// v: this lane's row.  tile: global address of the wave's first row (row0 * C floats into the array); rows: how many of the 64
// rows exist (the array's tail).  ALIGNED: `tile` is 16-byte aligned (true for any 16-byte aligned array when row0 % 64 == 0
// ... and C * 256 bytes is a multiple of 16: always).
template <int C, bool ALIGNED>
__device__ __forceinline__ void wave_store_rows(float* __restrict__ lds, const float (&v)[C], float* __restrict__ tile, const uint32_t rows,
                                                const uint32_t lane) {
    constexpr int P = row_pitch<C>();
#pragma unroll
    for (int c = 0; c < C; ++c) lds[lane * P + c] = v[c];
    __builtin_amdgcn_wave_barrier();
    const uint32_t total = rows * C;
    constexpr int kIters = (64 * C + 255) / 256;
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
        const uint32_t i = (it * 64 + lane) * 4;
        if (i >= total) continue;
        float f[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t e = i + k < 64 * C ? i + k : 64 * C - 1;
            f[k] = lds[(e / C) * P + e % C];
        }
        if (ALIGNED && i + 4 <= total) {
            *reinterpret_cast<float4*>(tile + i) = make_float4(f[0], f[1], f[2], f[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i + k < total) tile[i + k] = f[k];
        }
    }
    __builtin_amdgcn_wave_barrier();          // the slab may be rewritten by the wave's next tile
}

// the reverse: rows of a [B, C] array -> this lane's row (lanes beyond `rows` get zeros)
template <int C, bool ALIGNED>
__device__ __forceinline__ void wave_load_rows(float* __restrict__ lds, float (&v)[C], const float* __restrict__ tile, const uint32_t rows,
                                               const uint32_t lane) {
    constexpr int P = row_pitch<C>();
    const uint32_t total = rows * C;
    constexpr int kIters = (64 * C + 255) / 256;
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
        const uint32_t i = (it * 64 + lane) * 4;
        float f[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (i < total) {
            if (ALIGNED && i + 4 <= total) {
                const float4 q = *reinterpret_cast<const float4*>(tile + i);
                f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (i + k < total) f[k] = tile[i + k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t e = i + k;
            if (e < 64 * C) lds[(e / C) * P + e % C] = f[k];
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = lds[lane * P + c];
    __builtin_amdgcn_wave_barrier();
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace envidr
