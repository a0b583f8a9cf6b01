// Multiresolution feature-grid lookup core shared by hashencoder.hip, gridencoder.hip and the
// fused render kernel.
//
// Two interpolation flavours exist in the reference and they are NOT interchangeable
// (SURVEY.md App. B.4):
//   * "hash"  -- hashencoder/src/hashencoder.cu:103-254: smoothstep weights, pos = x*scale,
//                dense stride = resolution;
//   * "grid"  -- gridencoder/src/gridencoder.cu:75-223: linear weights, pos = x*scale + 0.5
//                (unless align_corners), dense stride = resolution+1 (resolution if aligned),
//                gridtype 1 ("tiled") never hashes.
//
// Per-level constants (scale, resolution, dense strides, hashed-or-not) depend only on
// (level, S, H, table size), so they are resolved once per launch on the host/scalar side into a
// LevelGeom instead of being re-derived by every lane as the reference does.  scale uses libm
// exp2f on the host so it is bit-identical to the CPU oracle on every machine.
#pragma once
#include "common.hip.h"
#include <math.h>
#include <type_traits>

namespace envidr {

constexpr int kMaxLevels = 32;
constexpr int kMaxDim = 5;

struct LevelScale {            // kernel argument block, filled by the launcher
    float scale[kMaxLevels];
    uint32_t resolution[kMaxLevels];
};

inline LevelScale make_level_scale(uint32_t L, float S, uint32_t H) {
    LevelScale ls;
    memset(&ls, 0, sizeof(ls));
    for (uint32_t l = 0; l < L && l < (uint32_t)kMaxLevels; ++l) {
        const float scale = exp2f(l * S) * H - 1.0f;              // hashencoder.cu:152
        ls.scale[l] = scale;
        ls.resolution[l] = (uint32_t)ceilf(scale) + 1;            // hashencoder.cu:153
    }
    return ls;
}

template <int D>
struct LevelGeom {
    uint32_t stride[D];   // dense strides (valid when !hashed)
    uint32_t size;        // rows in this level's table
    bool hashed;
    bool pow2;            // size is a power of two -> mask instead of modulo
};

// Mirrors get_grid_index's stride walk (hashencoder.cu:55-70 / gridencoder.cu:54-72) once per level.
template <int D>
__host__ __device__ __forceinline__ LevelGeom<D> make_level_geom(uint32_t size, uint32_t stride_step,
                                                                 bool allow_hash) {
    LevelGeom<D> g;
    g.size = size;
    g.pow2 = (size & (size - 1)) == 0;
    uint32_t stride = 1;
    int d = 0;
#pragma unroll
    for (; d < D; ++d) {
        if (!(stride <= size)) break;
        g.stride[d] = stride;
        stride *= stride_step;    // uint32 wrap-around kept, like the reference
    }
#pragma unroll
    for (int e = 0; e < D; ++e)
        if (e >= d) g.stride[e] = 0;   // dims the reference's loop never reached contribute nothing
    g.hashed = allow_hash && stride > size;
    return g;
}

template <int D>
__device__ __forceinline__ uint32_t cell_row(const LevelGeom<D>& g, const uint32_t (&p)[D]) {
    constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t idx = 0;
    if (g.hashed) {
#pragma unroll
        for (int d = 0; d < D; ++d) idx ^= p[d] * primes[d];
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d) idx += p[d] * g.stride[d];
    }
    return g.pow2 ? (idx & (g.size - 1)) : (idx % g.size);
}

template <int C>
struct Feat { float v[C]; };

// NT: non-temporal gathers (the fused renderer uses them so that the 48.8 MB table, which has no
// reuse at L2 scale, does not evict the < 1 MB of MLP weights every wave keeps re-streaming from L2)
template <int C, bool NT = false>
__device__ __forceinline__ Feat<C> load_row(const float* __restrict__ table, uint32_t row) {
    Feat<C> f;
    const float* p = table + (size_t)row * C;
    if constexpr (C == 2) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 t = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p)) : *reinterpret_cast<const f32x2*>(p);
        f.v[0] = t.x; f.v[1] = t.y;
    } else if constexpr (C == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        f.v[0] = t.x; f.v[1] = t.y; f.v[2] = t.z; f.v[3] = t.w;
    } else if constexpr (C == 8) {
        const float4 t0 = reinterpret_cast<const float4*>(p)[0], t1 = reinterpret_cast<const float4*>(p)[1];
        f.v[0] = t0.x; f.v[1] = t0.y; f.v[2] = t0.z; f.v[3] = t0.w;
        f.v[4] = t1.x; f.v[5] = t1.y; f.v[6] = t1.z; f.v[7] = t1.w;
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) f.v[c] = p[c];
    }
    return f;
}

// Evaluate one level at one point.
//   SMOOTH  : smoothstep weights (hash flavour) vs linear (grid flavour)
//   out[C]  : interpolated features
//   dydx    : [D][C] derivative w.r.t. the [0,1] input, written when WITH_GRAD
// All 2^D corner rows are gathered once (issued back to back so their latencies overlap) and
// reused for the derivative; the reference re-gathers them, values are identical.
template <int D, int C, bool SMOOTH, bool WITH_GRAD, bool NT = false>
__device__ __forceinline__ void eval_level(const float (&x)[D], const float* __restrict__ table,
                                           const LevelGeom<D>& g, float scale, float pos_offset, float (&out)[C],
                                           float (&dydx)[D][C]) {
    float w1[D];        // weight of the "+1" corner along d; the "+0" corner gets 1 - w1
    float dw[D];        // d(w1)/d(frac)
    uint32_t cell[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float p = x[d] * scale + pos_offset;
        const float fl = floorf(p);
        cell[d] = (uint32_t)fl;
        p -= (float)cell[d];
        if constexpr (SMOOTH) {
            dw[d] = 6 * p * (1.0f - p);
            w1[d] = p * p * (3.0f - 2.0f * p);
        } else {
            dw[d] = 1.0f;
            w1[d] = p;
        }
    }

    Feat<C> corner[1 << D];
#pragma unroll
    for (int i = 0; i < (1 << D); ++i) {
        uint32_t q[D];
#pragma unroll
        for (int d = 0; d < D; ++d) q[d] = cell[d] + ((i >> d) & 1);
        corner[i] = load_row<C, NT>(table, cell_row<D>(g, q));
    }

#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = 0;
#pragma unroll
    for (int i = 0; i < (1 << D); ++i) {
        float w = 1;
#pragma unroll
        for (int d = 0; d < D; ++d) w *= ((i >> d) & 1) ? w1[d] : 1 - w1[d];
#pragma unroll
        for (int c = 0; c < C; ++c) out[c] += w * corner[i].v[c];
    }

    if constexpr (WITH_GRAD) {
#pragma unroll
        for (int gd = 0; gd < D; ++gd) {
            float acc[C];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = 0;
#pragma unroll
            for (int j = 0; j < (1 << (D - 1)); ++j) {
                float w = scale;
                int lo = 0;   // corner index with bit gd cleared
#pragma unroll
                for (int nd = 0; nd < D - 1; ++nd) {
                    const int d = nd >= gd ? nd + 1 : nd;
                    const int bit = (j >> nd) & 1;
                    w *= bit ? w1[d] : 1 - w1[d];
                    lo |= bit << d;
                }
                const int hi = lo | (1 << gd);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if constexpr (SMOOTH) acc[c] += w * (corner[hi].v[c] - corner[lo].v[c]) * dw[gd];
                    else acc[c] += w * (corner[hi].v[c] - corner[lo].v[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) dydx[gd][c] = acc[c];
        }
    }
}

// Scatter with duplicates combined inside the wave.  The samples of a training batch are consecutive along their rays, so on the
// coarse levels the 64 points of a wave fall into a handful of cells and their atomics hit the SAME few addresses, which the L2
// atomic units serialise (measured on march_rays_train samples: the table scatter was slower than on uniformly random points, 13 G
// against 20 G atomics/s, and 33x the forward pass).  Lanes are grouped into runs of consecutive lanes with equal row; a segmented
// inclusive scan (six shuffle steps) leaves each run's sum in its last lane, which issues the one atomic.  `combine` is decided
// per wave and level from the first corner (wave-uniform): where neighbouring lanes never share a row -- the fine levels,
// random points -- every lane just issues its own atomics as before.  (Summation order changes; the scatter is order-dependent anyway.)
struct RunScatter {
    uint32_t lane;
    unsigned long long active;      // lanes that have something to add
    __device__ __forceinline__ RunScatter(bool on) : lane(__lane_id()), active(__ballot(on)) {}
    // heads of the runs for this corner's rows; returns this lane's run start and whether it is the run's last lane
    __device__ __forceinline__ void runs(uint32_t row, bool on, uint32_t& start, bool& tail, unsigned long long& heads) const {
        const uint32_t prev = __shfl_up(row, 1, 64);
        const bool prev_on = (active >> ((lane + 63u) & 63u)) & 1ull;
        const bool head = on && (lane == 0 || !prev_on || prev != row);
        heads = __ballot(head);
        const unsigned long long below = heads & ((2ull << lane) - 1ull);       // heads at or below this lane
        start = 63u - (uint32_t)__clzll(below | 1ull);
        const bool next_on = lane < 63u && ((active >> (lane + 1u)) & 1ull);
        const bool next_head = lane < 63u && ((heads >> (lane + 1u)) & 1ull);
        tail = on && (!next_on || next_head);
    }
    template <int C>
    __device__ __forceinline__ void add(float* __restrict__ t, uint32_t row, bool on, float (&v)[C], bool combine) const {
        if (combine) {
            uint32_t start; bool tail; unsigned long long heads;
            runs(row, on, start, tail, heads);
#pragma unroll
            for (uint32_t off = 1; off < 64; off <<= 1) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float up = __shfl_up(v[c], off, 64);
                    if (on && lane >= start + off) v[c] += up;
                }
            }
            on = tail;
        }
        if (on) {
#pragma unroll
            for (int c = 0; c < C; ++c) unsafeAtomicAdd(&t[(size_t)row * C + c], v[c]);
        }
    }
    // worth combining?  (at least a quarter of the active lanes share their row with the lane before them)
    __device__ __forceinline__ bool worth(uint32_t row, bool on) const {
        uint32_t start; bool tail; unsigned long long heads;
        runs(row, on, start, tail, heads);
        return 4 * __popcll(heads) <= 3 * __popcll(active);
    }
};


}  // namespace envidr
