// hashencoder operators for gfx950 (smoothstep multiresolution hash grid with analytic input
// derivative) -- replaces hashencoder/src/hashencoder.cu (hash_encode_forward :725,
// hash_encode_backward :762, hash_encode_second_backward :795; declarations hashencoder.h:13-15).
//
// Work decomposition: one lane per (point, level).  The forward kernels run their blocks in level-major order (the chip sweeps the
// levels together); the per-point table scatters still map blocks so that each XCD works on one level at a time (level = f(block % 8):
// a hashed level's 4 MiB table then meets its atomics in ONE L2).  With dy_dx a workgroup owns 128 points for all levels and assembles
// their dy_dx rows in LDS (k_hash_forward_rows); the input gradient moves the rows of 64 points per wave through LDS (input_rows.hip.h).
#include "grid_core.hip.h"
#include "input_rows.hip.h"
#include <algorithm>
#include <map>
#include <mutex>

using namespace envidr;

namespace {

constexpr uint32_t kXcds = 8;
__device__ constexpr uint32_t kCellPrimes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};   // grid_core.hip.h cell_row()
__host__ __device__ constexpr uint32_t ceil_div_u(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// block -> (level, chunk) such that blocks resident on one XCD (block % 8) sweep one level after
// another.  Returns false for padding blocks.
__device__ __forceinline__ bool xcd_level_chunk(uint32_t L, uint32_t chunks, uint32_t& level, uint32_t& chunk) {
    const uint32_t xcd = blockIdx.x % kXcds;
    const uint32_t j = blockIdx.x / kXcds;          // sequence number inside this XCD's queue
    const uint32_t k = j / chunks;                  // which of this XCD's levels
    level = xcd + kXcds * k;
    chunk = j - k * chunks;
    return level < L;
}
inline uint32_t xcd_grid_blocks(uint32_t L, uint32_t chunks) { return kXcds * ceil_div(L, kXcds) * chunks; }

template <int D>
__device__ __forceinline__ bool load_point(const float* __restrict__ inputs, uint32_t b, float (&x)[D]) {
    bool inside = true;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        x[d] = inputs[(size_t)b * D + d];
        if (x[d] < 0 || x[d] > 1) inside = false;
    }
    return inside;
}

// ------------------------------------------------------------------------------------------
// forward (+ optional dy_dx)
// ------------------------------------------------------------------------------------------
template <int D, int C, bool GRAD>
__global__ void __launch_bounds__(kBlock) k_hash_forward(const float* __restrict__ inputs,
                                                         const float* __restrict__ embeddings,
                                                         const int32_t* __restrict__ offsets,
                                                         float* __restrict__ outputs, uint32_t B, uint32_t L,
                                                         LevelScale ls, uint32_t chunks, float* __restrict__ dy_dx) {
    // blocks in level-major order (all chunks of level 0, then level 1 ...): the chip sweeps the levels together.  Rounds 1-4 pinned levels
    // to XCDs here (block % 8 -> level) so that a level's 4 MiB table stays in ONE L2; timed against the reference's own kernel compiled
    // for this GPU (tools/ops_bench.py) that was the slower choice on every input order -- the XCDs finish their levels at different times
    // (coarse levels hit, fine ones miss) and the others' L2s sit idle: 3.87 -> 3.50 ms on random points, 2.54 -> 1.80 ms on a frame's
    // ray-major samples, 3.15 -> 2.64 ms sample-major.
    const uint32_t level = blockIdx.x / chunks, chunk = blockIdx.x - level * chunks;
    const uint32_t b = chunk * blockDim.x + threadIdx.x;
    if (b >= B) return;

    const uint32_t row0 = (uint32_t)offsets[level];
    const uint32_t size = (uint32_t)offsets[level + 1] - row0;
    const LevelGeom<D> g = make_level_geom<D>(size, ls.resolution[level], /*allow_hash=*/true);

    float x[D], out[C], grad[D][C];
    const bool inside = load_point<D>(inputs, b, x);
    if (inside) {
        eval_level<D, C, /*SMOOTH=*/true, GRAD>(x, embeddings + (size_t)row0 * C, g, ls.scale[level], 0.0f, out, grad);
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) out[c] = 0;
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int c = 0; c < C; ++c) grad[d][c] = 0;
    }

    float* o = outputs + ((size_t)level * B + b) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = out[c];
    if constexpr (GRAD) {
        float* g_out = dy_dx + ((size_t)b * L + level) * (D * C);
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int c = 0; c < C; ++c) g_out[d * C + c] = grad[d][c];
    }
}

// forward + dy_dx, rows assembled in LDS (round 5).  With one level per workgroup (above) a point's 16 x 24-byte pieces of its dy_dx row are
// written by 16 different workgroups on different XCDs at different times: 3.1 GB of 24-byte stores at a 384-byte stride cost more than the
// gathers (6.5 ms against 2.5 without dy_dx, on a frame's own samples).  Here a workgroup owns 128 points for ALL levels -- thread (p, h) walks
// the levels of parity h -- parks the derivatives in LDS (row pitch odd: conflict-free) and writes the 128 rows, which are one contiguous range
// of dy_dx, as full lines.  Same arithmetic per (point, level) as k_hash_forward: same bits.
constexpr uint32_t kRowsPoints = 128;
template <int D, int C>
__global__ void __launch_bounds__(kBlock) k_hash_forward_rows(const float* __restrict__ inputs, const float* __restrict__ embeddings,
                                                              const int32_t* __restrict__ offsets, float* __restrict__ outputs, uint32_t B,
                                                              uint32_t L, LevelScale ls, float* __restrict__ dy_dx, uint32_t row_magic) {
    extern __shared__ float s_rows[];                       // [kRowsPoints][pitch]
    const uint32_t row_floats = L * D * C, pitch = row_floats | 1u;
    const uint32_t p = threadIdx.x & (kRowsPoints - 1), par = threadIdx.x >> 7;
    const uint32_t b0 = blockIdx.x * kRowsPoints, b = b0 + p;
    if (b < B) {
        float x[D];
        const bool inside = load_point<D>(inputs, b, x);
        for (uint32_t level = par; level < L; level += 2) {
            const uint32_t row0 = (uint32_t)offsets[level];
            const uint32_t size = (uint32_t)offsets[level + 1] - row0;
            const LevelGeom<D> g = make_level_geom<D>(size, ls.resolution[level], /*allow_hash=*/true);
            float out[C], grad[D][C];
            if (inside) {
                eval_level<D, C, /*SMOOTH=*/true, true>(x, embeddings + (size_t)row0 * C, g, ls.scale[level], 0.0f, out, grad);
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) out[c] = 0;
#pragma unroll
                for (int d = 0; d < D; ++d)
#pragma unroll
                    for (int c = 0; c < C; ++c) grad[d][c] = 0;
            }
            float* o = outputs + ((size_t)level * B + b) * C;
#pragma unroll
            for (int c = 0; c < C; ++c) o[c] = out[c];
            float* r = s_rows + p * pitch + level * (D * C);
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int c = 0; c < C; ++c) r[d * C + c] = grad[d][c];
        }
    }
    __syncthreads();
    const uint32_t points = min(kRowsPoints, B - b0);
    float* dst = dy_dx + (size_t)b0 * row_floats;
    for (uint32_t e = threadIdx.x; e < points * row_floats; e += kBlock) {
        const uint32_t q = __umulhi(e, row_magic), col = e - q * row_floats;          // e / row_floats (row_division_magic)
        dst[e] = s_rows[q * pitch + col];
    }
}

// ------------------------------------------------------------------------------------------
// backward w.r.t. the table: scatter w * grad into the 2^D corner rows (hardware fp32 atomics)
// ------------------------------------------------------------------------------------------
template <int D, int C>
__global__ void __launch_bounds__(kBlock) k_hash_backward_table(const float* __restrict__ grad,
                                                                const float* __restrict__ inputs,
                                                                const int32_t* __restrict__ offsets,
                                                                float* __restrict__ grad_table, uint32_t B, uint32_t L,
                                                                LevelScale ls, uint32_t chunks) {
    uint32_t level, chunk;
    if (!xcd_level_chunk(L, chunks, level, chunk)) return;
    const uint32_t b = chunk * blockDim.x + threadIdx.x;
    // (no early exit per lane: the lanes of a wave combine their adds, RunScatter)
    float x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = 0.5f;
    const bool on = b < B && load_point<D>(inputs, b, x);   // outside the cube: nothing to add (the table gradient starts at zero)
    if (!on) {
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = 0.5f;
    }

    const uint32_t row0 = (uint32_t)offsets[level];
    const uint32_t size = (uint32_t)offsets[level + 1] - row0;
    const LevelGeom<D> g = make_level_geom<D>(size, ls.resolution[level], true);
    const float scale = ls.scale[level];

    float w1[D];
    uint32_t cell[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float p = x[d] * scale;
        cell[d] = (uint32_t)floorf(p);
        p -= (float)cell[d];
        w1[d] = p * p * (3.0f - 2.0f * p);
    }
    float gcur[C];
#pragma unroll
    for (int c = 0; c < C; ++c) gcur[c] = on ? grad[((size_t)level * B + b) * C + c] : 0.0f;

    float* t = grad_table + (size_t)row0 * C;
    const RunScatter rs(on);
    bool combine = false;
#pragma unroll
    for (int i = 0; i < (1 << D); ++i) {
        float w = 1;
        uint32_t q[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int bit = (i >> d) & 1;
            w *= bit ? w1[d] : 1 - w1[d];
            q[d] = cell[d] + bit;
        }
        const uint32_t row = cell_row<D>(g, q);
        if (i == 0) combine = rs.worth(row, on);
        float v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = w * gcur[c];
        rs.template add<C>(t, row, on, v, combine);
    }
}

// grad_inputs[b,d] = sum_{l,c} grad[l,b,c] * dy_dx[b,l,d,c]      (hashencoder.cu:346-372): input_rows.hip.h

// grad_grad[l,b,c] = sum_d ggx[b,d] * dy_dx[b,l,d,c]              (hashencoder.cu:375-428)
template <int D, int C>
__global__ void __launch_bounds__(kBlock) k_second_backward_grad(const float* __restrict__ ggx,
                                                                 const float* __restrict__ dy_dx,
                                                                 float* __restrict__ grad_grad, uint32_t B,
                                                                 uint32_t L) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t level = blockIdx.y;
    if (b >= B) return;
    const float* j = dy_dx + ((size_t)b * L + level) * (D * C);
    float r[C];
#pragma unroll
    for (int c = 0; c < C; ++c) r[c] = 0;
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int c = 0; c < C; ++c) r[c] += ggx[(size_t)b * D + d] * j[d * C + c];
#pragma unroll
    for (int c = 0; c < C; ++c) grad_grad[((size_t)level * B + b) * C + c] = r[c];
}

// table second gradient: +/- w * grad * ggx[gd] * smoothstep'(frac_gd) on the corner pairs
// along gd, accumulated per corner in registers then scattered      (hashencoder.cu:431-595)
template <int D, int C>
__global__ void __launch_bounds__(kBlock) k_second_backward_table(const float* __restrict__ grad,
                                                                  const float* __restrict__ inputs,
                                                                  const int32_t* __restrict__ offsets,
                                                                  const float* __restrict__ ggx,
                                                                  float* __restrict__ grad2_table, uint32_t B,
                                                                  uint32_t L, LevelScale ls, uint32_t chunks) {
    uint32_t level, chunk;
    if (!xcd_level_chunk(L, chunks, level, chunk)) return;
    const uint32_t b = chunk * blockDim.x + threadIdx.x;
    float x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = 0.5f;
    const bool on = b < B && load_point<D>(inputs, b, x);          // (no early exit per lane: RunScatter combines across the wave)
    if (!on) {
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = 0.5f;
    }

    const uint32_t row0 = (uint32_t)offsets[level];
    const uint32_t size = (uint32_t)offsets[level + 1] - row0;
    const LevelGeom<D> g = make_level_geom<D>(size, ls.resolution[level], true);
    const float scale = ls.scale[level];

    float w1[D], dw[D];
    uint32_t cell[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float p = x[d] * scale;
        cell[d] = (uint32_t)floorf(p);
        p -= (float)cell[d];
        dw[d] = 6 * p * (1.0f - p);
        w1[d] = p * p * (3.0f - 2.0f * p);
    }
    float gcur[C], gg[D];
#pragma unroll
    for (int c = 0; c < C; ++c) gcur[c] = on ? grad[((size_t)level * B + b) * C + c] : 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) gg[d] = on ? ggx[(size_t)b * D + d] : 0.0f;

    float corner[1 << D][C];
#pragma unroll
    for (int i = 0; i < (1 << D); ++i)
#pragma unroll
        for (int c = 0; c < C; ++c) corner[i][c] = 0;

#pragma unroll
    for (int gd = 0; gd < D; ++gd) {
#pragma unroll
        for (int j = 0; j < (1 << (D - 1)); ++j) {
            float w = scale;
            int lo = 0;
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
                const float v = w * gcur[c] * gg[gd] * dw[gd];
                corner[hi][c] += v;
                corner[lo][c] -= v;
            }
        }
    }

    float* t = grad2_table + (size_t)row0 * C;
    const RunScatter rs(on);
    bool combine = false;
#pragma unroll
    for (int i = 0; i < (1 << D); ++i) {
        uint32_t q[D];
#pragma unroll
        for (int d = 0; d < D; ++d) q[d] = cell[d] + ((i >> d) & 1);
        const uint32_t row = cell_row<D>(g, q);
        if (i == 0) combine = rs.worth(row, on);
        rs.template add<C>(t, row, on, corner[i], combine);
    }
}

// ------------------------------------------------------------------------------------------
// Table gradients of LARGE batches: range-owned accumulation in LDS.
//
// Global fp32 atomics run at ~20 G/s on this chip whatever their addresses and however many XCDs issue them (measured:
// 6.2 ms per level and 7.7 M points, linear in the number of levels although every level runs on its own XCD) -- 2 x 10^9 of
// them are the 98 ms of the scatter kernels above.  A level's table has far fewer rows (<= 2^19) than a large batch has
// contributions (8 B), so the sums are formed in LDS instead: a workgroup OWNS a range of kLdsRows rows of one level (128 KiB
// of accumulators) and adds the corner contributions that fall into its range with LDS atomics; at the end the range is
// flushed with one global atomic per non-zero entry.  The 32 CUs of an XCD take the 32 ranges of one level at the same time.
//
// Which points does a range owner look at?  A hashed level scatters a point's 2^D corners over the whole table: a given
// range (1/32 of the rows) holds a corner of ~22 % of the points, and a first version in which every owner recomputed
// every point's corner rows to find out spent its time in that index arithmetic (1.1e10 vector instructions, 20.9 ms for
// 7.7 M points).  So a pre-pass (k_table_range_masks, 0.34 ms) forms the rows once per (point, level) and leaves a 32-bit
// word: bit r = "a corner lies in a range owned by slot r".  An owner streams the words of its points (16 bytes per lane
// and load), ballots "selected", and every lane finds the point it is to work on from the ballots alone (nth_set_bit), so
// the index and weight arithmetic runs on the selected points only: 1.9e9 vector instructions, 12.0 ms.
// What bounds it now (tools/probe/lds_atomic_probe.hip, scatter_timers / scatter_phases variants): an fp32 LDS atomic
// takes 1.9 cycles of the CU's LDS pipe PER LANE (120 for a full wave, whatever the banks), i.e. 2 x 10^9 lane-atomics =
// 6.0 ms if the pipe never idled; it is busy 48 % of the kernel -- the waves of a workgroup spend their time one behind the
// other in the atomic section while the pipe serves them in turn, and the other half of the time all of them compute
// (starting the waves of a workgroup out of step changes nothing).
// ------------------------------------------------------------------------------------------
constexpr uint32_t kLdsScatterThreads = 1024;
constexpr uint32_t kLdsScatterWaves = kLdsScatterThreads / 64;
constexpr uint32_t kLdsScatterFloats = 32768;          // 128 KiB of accumulators per workgroup
constexpr uint32_t kLdsScatterSlots = 32;              // range slots per (level, part) group = CUs per XCD
constexpr uint32_t kLdsBatch = 2;                      // mask loads (256 points each) a wave works on side by side: points per lane in flight
constexpr uint32_t kMaskGrain = 256;                   // points per wave and mask load (4 per lane); point parts start at multiples of it

// Work distribution.  The launch is 256 persistent workgroups: 8 XCDs x 32 slots (blockIdx % 8 is the XCD).  In step n the 32
// slots of XCD x take "super-group" 8 n + x together.  A super-group belongs to one level and holds k = 32 / ranges groups of
// `ranges` slots each (ranges = row ranges of the level, at most 32 per pass); a group is one part of the level's points, its
// slots that part's row ranges -- so the slots of a group stream the same points at the same time (out of their XCD's L2), and
// a level with few ranges (the dense ones: 1, 1, 2, 5, 13) still keeps ~kLdsBlocksPerLevel workgroups busy instead of doing all
// its LDS atomics on a handful (level 0 on three workgroups took 51 ms).  The plan is recomputed from `offsets` by every
// workgroup (scalar work; the sizes live on the device).
constexpr uint32_t kLdsBlocksPerLevel = 288;
struct LevelPlan { uint32_t ranges, per_sg, parts, sgs; };
__device__ __forceinline__ LevelPlan level_plan(uint32_t size, uint32_t rows_per_range) {
    LevelPlan p;
    p.ranges = min(kLdsScatterSlots, max(1u, ceil_div_u(size, rows_per_range)));
    p.per_sg = kLdsScatterSlots / p.ranges;
    p.parts = ceil_div_u(ceil_div_u(kLdsBlocksPerLevel, p.ranges), p.per_sg) * p.per_sg;
    p.sgs = p.parts / p.per_sg;
    return p;
}

// position of the r-th (0-based) set bit of a wave-uniform 64-bit mask, r < popcount(mask): five popcount steps per lane
__device__ __forceinline__ uint32_t nth_set_bit(unsigned long long mask, uint32_t r) {
    const uint32_t lo = (uint32_t)mask, hi = (uint32_t)(mask >> 32);
    const uint32_t pl = (uint32_t)__popc(lo);
    const bool up = r >= pl;
    uint32_t w = up ? hi : lo;
    r = up ? r - pl : r;
    uint32_t pos = up ? 32u : 0u;
#pragma unroll
    for (uint32_t width = 16; width >= 1; width >>= 1) {
        const uint32_t c = (uint32_t)__popc(w & ((1u << width) - 1u));
        const bool go = r >= c;
        r = go ? r - c : r;
        w = go ? (w >> width) : w;
        pos += go ? width : 0u;
    }
    return pos;
}

// a point's position inside its cell and the per-dimension products of cell_row(): (c + 1) * m is c * m + m in uint32
// arithmetic, so the rows of the 2^D corners are D multiplies (quarter-rate instructions), not D per corner
// (lo / hi in separate arrays: a run-time choice between term[d][0] and term[d][1] of ONE array becomes an indexed access, i.e. the
//  array goes to scratch memory and every round of the atomic loop waits for a scratch load and everything else in flight)
template <int D> struct CornerTerms { float w1[D], dw[D]; uint32_t lo[D], hi[D]; };
template <int D>
__device__ __forceinline__ void corner_terms(const LevelGeom<D>& g, float scale, const float (&x)[D], CornerTerms<D>& t) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float p = x[d] * scale;
        const uint32_t cell = (uint32_t)floorf(p);
        p -= (float)cell;
        t.dw[d] = 6 * p * (1.0f - p);
        t.w1[d] = p * p * (3.0f - 2.0f * p);
        const uint32_t m = g.hashed ? kCellPrimes[d] : g.stride[d];
        t.lo[d] = cell * m;
        t.hi[d] = t.lo[d] + m;
    }
}
template <int D>
__device__ __forceinline__ uint32_t corner_row(const LevelGeom<D>& g, const CornerTerms<D>& t, int i) {
    uint32_t idx = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) { const uint32_t tm = ((i >> d) & 1) ? t.hi[d] : t.lo[d]; idx = g.hashed ? (idx ^ tm) : (idx + tm); }
    if (g.pow2) idx &= g.size - 1;
    else if (idx >= g.size) idx %= g.size;          // dense levels: only the cube's far faces reach past the last row
    return idx;
}

// masks[level * pitch + b]: bit ((row / rows_per_range) % 32) for the 2^D corner rows of point b (0 outside the cube)
template <int D>
__global__ void __launch_bounds__(kBlock) k_table_range_masks(const float* __restrict__ inputs, const int32_t* __restrict__ offsets,
                                                              uint32_t* __restrict__ masks, uint32_t B, uint32_t pitch, uint32_t log2_rows,
                                                              LevelScale ls) {
    const uint32_t level = blockIdx.y, b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    const uint32_t row0 = (uint32_t)offsets[level];
    const uint32_t size = (uint32_t)offsets[level + 1] - row0;
    const LevelGeom<D> g = make_level_geom<D>(size, ls.resolution[level], true);
    float x[D];
    uint32_t mask = 0;
    if (load_point<D>(inputs, b, x)) {
        CornerTerms<D> t;
        corner_terms<D>(g, ls.scale[level], x, t);
#pragma unroll
        for (int i = 0; i < (1 << D); ++i) mask |= 1u << ((corner_row<D>(g, t, i) >> log2_rows) & 31u);
    }
    masks[(size_t)level * pitch + b] = mask;
}

template <int D, int C, bool SECOND>
__global__ void __launch_bounds__(kLdsScatterThreads) k_table_scatter_lds(const float* __restrict__ grad, const float* __restrict__ inputs,
                                                                          const int32_t* __restrict__ offsets, const float* __restrict__ ggx,
                                                                          const uint32_t* __restrict__ masks, uint32_t pitch,
                                                                          float* __restrict__ grad_table, uint32_t B, uint32_t L, LevelScale ls) {
    constexpr uint32_t kRows = kLdsScatterFloats / C;
    __shared__ float s_acc[kLdsScatterFloats];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // block -> (group, range slot): the blocks of one XCD (blockIdx % 8) take the 32 ranges of one group at a time
    const uint32_t xcd = blockIdx.x % kXcds, slot = (blockIdx.x / kXcds) % kLdsScatterSlots;
  for (uint32_t step = 0;; ++step) {
    const uint32_t sg = step * kXcds + xcd;
    uint32_t level = 0, first = 0;
    LevelPlan plan = {};
    for (; level < L; ++level) {
        plan = level_plan((uint32_t)(offsets[level + 1] - offsets[level]), kRows);
        if (sg < first + plan.sgs) break;
        first += plan.sgs;
    }
    if (level >= L) break;                                   // past the last super-group (uniform over the workgroup)
    const uint32_t sub = slot / plan.ranges;
    if (sub >= plan.per_sg) continue;                        // 32 is not a multiple of this level's range count: a spare slot
    const uint32_t parts = plan.parts, part = (sg - first) * plan.per_sg + sub, first_range = slot - sub * plan.ranges;
    const uint32_t row0 = (uint32_t)offsets[level];
    const uint32_t size = (uint32_t)offsets[level + 1] - row0;
    const uint32_t ranges = ceil_div_u(size, kRows);
    // a level with more than 32 ranges (a table beyond 2^19 x 2 floats per level) wraps around: a slot owns ranges r, r + 32, ...
    // (all of them share the slot's mask bit)
    const LevelGeom<D> g = make_level_geom<D>(size, ls.resolution[level], true);
    const float scale = ls.scale[level];
    // point parts start at multiples of the mask grain (16-byte aligned mask loads)
    const uint32_t p0 = (uint32_t)(((unsigned long long)B * part) / parts) / kMaskGrain * kMaskGrain;
    const uint32_t p1 = part + 1 == parts ? B : (uint32_t)(((unsigned long long)B * (part + 1)) / parts) / kMaskGrain * kMaskGrain;
    const uint32_t* mrow = masks + (size_t)level * pitch;
    for (uint32_t range = first_range; range < ranges; range += kLdsScatterSlots) {
        const uint32_t base = range * kRows;
        for (uint32_t i = threadIdx.x; i < kLdsScatterFloats; i += kLdsScatterThreads) s_acc[i] = 0.0f;
        __syncthreads();

        // A batch is kLdsBatch points per lane: loads, index and weight arithmetic, LDS atomics.  NOTHING else here may touch LDS:
        // the CU's LDS pipe takes ~1.9 cycles per lane of an fp32 atomic (120 per full wave) and serves plain LDS reads and writes of
        // ANY wave only behind the atomics in flight (tools/probe/lds_atomic_probe.hip: a wave of sparse ds_writes beside atomic waves
        // finishes when the atomics do) -- with the selected points queued through LDS the kernel took the SUM of its atomics (7.4 ms)
        // and of everything else (5.4 ms); vector-ALU work and global loads do overlap the atomics.
        struct Batch { uint32_t b[kLdsBatch]; bool on[kLdsBatch]; float xs[kLdsBatch][D], gs[kLdsBatch][C], ggs[kLdsBatch][D]; };
        struct Prepared { CornerTerms<D> t; uint32_t pending; float gcur[C]; float corner[SECOND ? (1 << D) : 1][C]; };
        auto issue_loads = [&](Batch& q) {
#pragma unroll
            for (uint32_t u = 0; u < kLdsBatch; ++u) {
                const uint32_t b = q.b[u];
#pragma unroll
                for (int d = 0; d < D; ++d) q.xs[u][d] = inputs[(size_t)b * D + d];
#pragma unroll
                for (int c = 0; c < C; ++c) q.gs[u][c] = grad[((size_t)level * B + b) * C + c];
                if constexpr (SECOND) {
#pragma unroll
                    for (int d = 0; d < D; ++d) q.ggs[u][d] = ggx[(size_t)b * D + d];
                }
            }
        };
        auto prepare = [&](const Batch& q, uint32_t u, Prepared& r) {
            const bool on = q.on[u];          // (a queued point is inside the cube: its mask would be 0 otherwise)
            float x[D];
#pragma unroll
            for (int d = 0; d < D; ++d) x[d] = on ? q.xs[u][d] : 0.5f;
            corner_terms<D>(g, scale, x, r.t);
            r.pending = 0;
#pragma unroll
            for (int i = 0; i < (1 << D); ++i) r.pending |= (on && corner_row<D>(g, r.t, i) - base < kRows) ? (1u << i) : 0u;
#pragma unroll
            for (int c = 0; c < C; ++c) r.gcur[c] = on ? q.gs[u][c] : 0.0f;
            if constexpr (SECOND) {
                // +/- w * grad * ggx[gd] * smoothstep'(frac_gd) on the corner pairs along gd (k_second_backward_table's statements)
                float gg[D];
#pragma unroll
                for (int d = 0; d < D; ++d) gg[d] = on ? q.ggs[u][d] : 0.0f;
#pragma unroll
                for (int i = 0; i < (1 << D); ++i)
#pragma unroll
                    for (int c = 0; c < C; ++c) r.corner[i][c] = 0;
#pragma unroll
                for (int gd = 0; gd < D; ++gd) {
#pragma unroll
                    for (int jj = 0; jj < (1 << (D - 1)); ++jj) {
                        float w = scale;
                        int lo = 0;
#pragma unroll
                        for (int nd = 0; nd < D - 1; ++nd) {
                            const int d = nd >= gd ? nd + 1 : nd;
                            const int bit = (jj >> nd) & 1;
                            w *= bit ? r.t.w1[d] : 1 - r.t.w1[d];
                            lo |= bit << d;
                        }
                        const int hi = lo | (1 << gd);
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            const float v = w * r.gcur[c] * gg[gd] * r.t.dw[gd];
                            r.corner[hi][c] += v;
                            r.corner[lo][c] -= v;
                        }
                    }
                }
            }
        };
        // Every lane picks ITS next in-range corner and the wave issues one atomic per channel and round; a queued point has at
        // least one corner in range, so the first round is a full wave; ~1.1 corners per point are in range on a hashed level.
        auto add = [&](Prepared& r) {
            uint32_t pending = r.pending;
            while (__ballot(pending != 0)) {
                const int pick = pending ? __ffs((int)pending) - 1 : 0;
                // the picked corner's row and weight from its bits: D selects each, instead of 2^D-way select chains
                uint32_t idx = 0;
                float w = 1;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const bool bit = (pick >> d) & 1;
                    const uint32_t tm = bit ? r.t.hi[d] : r.t.lo[d];
                    idx = g.hashed ? (idx ^ tm) : (idx + tm);
                    w *= bit ? r.t.w1[d] : 1 - r.t.w1[d];
                }
                if (g.pow2) idx &= g.size - 1;
                else if (idx >= g.size) idx %= g.size;
                const uint32_t at = idx - base;
                float v[C];
                if constexpr (SECOND) {
#pragma unroll
                    for (int c = 0; c < C; ++c) v[c] = r.corner[0][c];
#pragma unroll
                    for (int i = 1; i < (1 << D); ++i) {
                        if (pick == i) {
#pragma unroll
                            for (int c = 0; c < C; ++c) v[c] = r.corner[i][c];
                        }
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < C; ++c) v[c] = w * r.gcur[c];
                }
                bool mine = pending != 0;
                // Dense levels: consecutive points of a frame or a training batch (consecutive samples of a ray, neighbouring rays) sit in the same
                // cell, so neighbouring lanes add to the SAME row -- whichever corner each of them picked.  Runs of equal rows are summed in the wave
                // (segmented scan, grid_core.hip.h RunScatter) and the run's last lane issues the one atomic: the LDS atomic unit takes its 1.9
                // cycles per LANE, and the few ranges of a dense level that hold the scene get most of its points.  Decided per round from the
                // rows themselves (wave-uniform); on the hashed levels neighbours never share a row and nothing of this runs.
                if (!g.hashed) {
                    const RunScatter rs(mine);
                    uint32_t start; bool tail; unsigned long long heads;
                    rs.runs(at, mine, start, tail, heads);
                    if (4 * __popcll(heads) <= 3 * __popcll(rs.active)) {
#pragma unroll
                        for (uint32_t off = 1; off < 64; off <<= 1) {
#pragma unroll
                            for (int c = 0; c < C; ++c) {
                                const float up = __shfl_up(v[c], off, 64);
                                if (mine && lane >= start + off) v[c] += up;
                            }
                        }
                        mine = tail;
                    }
                }
                if (mine) {
#pragma unroll
                    for (int c = 0; c < C; ++c) __hip_atomic_fetch_add(&s_acc[c * kRows + at], v[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                pending &= pending - 1;
            }
        };

        // Which lane works on which selected point is computed from the ballots alone: the ballot of "point 4 l + k of this load is
        // selected" (k = 0 .. 3) is wave-uniform, so lane j can find the j-th selected point by itself (rank -> position of the
        // rank-th set bit: five popcount steps) -- no queue, no cross-lane traffic.  A load of 256 points selects ~56 on a hashed
        // level, i.e. one round with 7/8 of the lanes busy; kLdsBatch loads are worked on side by side (their point loads in
        // flight together).
        const uint32_t bit = first_range;                 // == range % 32
        constexpr uint32_t kStride = kLdsScatterWaves * kLdsBatch * kMaskGrain;
        auto load_masks = [&](uint32_t c0, uint4 (&m)[kLdsBatch]) {
#pragma unroll
            for (uint32_t u = 0; u < kLdsBatch; ++u) {
                const uint32_t b = c0 + u * kMaskGrain + lane * 4u;
                m[u] = make_uint4(0, 0, 0, 0);
                if (b < p1) m[u] = *reinterpret_cast<const uint4*>(mrow + b);          // (the row's tail is padded: pitch)
            }
        };
        uint4 m_next[kLdsBatch];
        load_masks(p0 + wave * kLdsBatch * kMaskGrain, m_next);
        for (uint32_t c0 = p0 + wave * kLdsBatch * kMaskGrain; c0 < p1; c0 += kStride) {
            uint4 m[kLdsBatch];
#pragma unroll
            for (uint32_t u = 0; u < kLdsBatch; ++u) m[u] = m_next[u];
            load_masks(c0 + kStride, m_next);          // (the next words are requested before these points are worked on)
            unsigned long long bal[kLdsBatch][4];
            uint32_t cum[kLdsBatch][4], most = 0;      // cum[u][k]: selected points of load u with a component below k
#pragma unroll
            for (uint32_t u = 0; u < kLdsBatch; ++u) {
                const uint32_t b = c0 + u * kMaskGrain + lane * 4u;
                const uint32_t mk[4] = {m[u].x, m[u].y, m[u].z, m[u].w};
                uint32_t n = 0;
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    bal[u][k] = __ballot(((mk[k] >> bit) & 1u) != 0 && b + k < p1);
                    cum[u][k] = n;
                    n += (uint32_t)__popcll(bal[u][k]);
                }
                most = max(most, n);
                cum[u][0] = n;                         // (cum[u][0] is 0 by construction: the slot carries the load's total instead)
            }
            for (uint32_t r0 = 0; r0 < most; r0 += 64u) {
                Batch q;
                const uint32_t j = r0 + lane;
#pragma unroll
                for (uint32_t u = 0; u < kLdsBatch; ++u) {
                    q.on[u] = j < cum[u][0];
                    const uint32_t k = (j >= cum[u][1] ? 1u : 0u) + (j >= cum[u][2] ? 1u : 0u) + (j >= cum[u][3] ? 1u : 0u);
                    const unsigned long long mask = k == 0 ? bal[u][0] : k == 1 ? bal[u][1] : k == 2 ? bal[u][2] : bal[u][3];
                    const uint32_t below = k == 0 ? 0u : k == 1 ? cum[u][1] : k == 2 ? cum[u][2] : cum[u][3];
                    const uint32_t src = q.on[u] ? nth_set_bit(mask, j - below) : 0u;
                    q.b[u] = q.on[u] ? c0 + u * kMaskGrain + src * 4u + k : p0;
                }
                issue_loads(q);
                Prepared prep[kLdsBatch];
#pragma unroll
                for (uint32_t u = 0; u < kLdsBatch; ++u) prepare(q, u, prep[u]);
#pragma unroll
                for (uint32_t u = 0; u < kLdsBatch; ++u) add(prep[u]);
            }
        }
        __syncthreads();
        float* t = grad_table + ((size_t)row0 + base) * C;
        const uint32_t valid = min(kRows, size - base) * C;
        for (uint32_t i = threadIdx.x; i < valid; i += kLdsScatterThreads) {
            const float v = s_acc[(i % C) * kRows + i / C];          // accumulators are channel-major: an atomic's 64 rows spread over all banks
            if (v != 0.0f) unsafeAtomicAdd(&t[i], v);
        }
        // (the accumulators may be zeroed again once every wave has READ them: wait for the LDS reads only, not for the global atomics)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
}

// Scratch for the range masks (4 bytes per point and level).  Batches of up to kSmallMaskBytes of masks (a training step's: 146 k samples x 16
// levels = 9 MiB) use a buffer kept per (device, stream) -- at most kSmallMaskBytes each, allocated once: `hipMallocAsync` / `hipFreeAsync` per call
// made the HOST wait about as long as the kernels run (0.5 ms per scatter measured, tools/probe/host_call_cost.py: twice a training step that is
// itself host-bound).  Larger batches take a stream-ordered allocation from the device's memory pool, released behind the kernels that read it: there
// the wait is small beside 10 ms of kernels, nothing outlives the call, and the pool hands the bytes back (round 4 kept an unbounded per-stream
// hipMalloc that was never freed: 0.5 GiB at 8 M points x 16 levels, outside torch's allocator).
constexpr size_t kSmallMaskBytes = 64u << 20;
// The kept buffer is as large as the largest request seen on its (device, stream) -- rounded up to a power of two from 1 MiB, at most
// kSmallMaskBytes -- not a flat 64 MiB: a training run that scatters 9 MiB of masks pins 16 MiB outside torch's allocator.  It grows by
// hipFree + hipMalloc (hipFree waits for the device: the kernels reading the old buffer are done).  envidr_release_scratch() hands all of it
// back (entries of streams that no longer exist included).
struct SmallMaskScratch { uint32_t* ptr = nullptr; size_t bytes = 0; };
static std::mutex g_small_mask_mutex;
static std::map<std::pair<int, hipStream_t>, SmallMaskScratch> g_small_masks;
static uint32_t* small_mask_scratch(hipStream_t s, size_t need) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lock(g_small_mask_mutex);
    SmallMaskScratch& e = g_small_masks[{dev, s}];
    if (e.bytes < need) {
        size_t want = 1u << 20;
        while (want < need) want <<= 1;
        if (e.ptr) (void)hipFree(e.ptr);
        e.ptr = nullptr;
        e.bytes = 0;
        if (hipMalloc(reinterpret_cast<void**>(&e.ptr), want) != hipSuccess) { (void)hipGetLastError(); e.ptr = nullptr; return nullptr; }
        e.bytes = want;
    }
    return e.ptr;
}
size_t release_small_mask_scratch() {
    std::lock_guard<std::mutex> lock(g_small_mask_mutex);
    size_t freed = 0;
    int keep = 0;
    (void)hipGetDevice(&keep);
    for (auto& kv : g_small_masks) {
        if (!kv.second.ptr) continue;
        (void)hipSetDevice(kv.first.first);
        if (hipFree(kv.second.ptr) == hipSuccess) freed += kv.second.bytes;
        else (void)hipGetLastError();
    }
    g_small_masks.clear();
    (void)hipSetDevice(keep);
    return freed;
}
// pre-pass + range owners
template <int D, int C, bool SECOND>
static int launch_table_scatter_lds(const float* grad, const float* inputs, const int32_t* offsets, const float* ggx, float* grad_table, uint32_t B,
                                    uint32_t L, const LevelScale& ls, hipStream_t s) {
    constexpr uint32_t kRows = kLdsScatterFloats / C;
    static_assert((kRows & (kRows - 1)) == 0, "rows per range: a power of two");
    uint32_t log2_rows = 0;
    while ((1u << log2_rows) < kRows) ++log2_rows;
    const uint32_t pitch = (B + 3u) / 4u * 4u;
    const size_t mask_bytes = ((size_t)L * pitch + kMaskGrain) * 4;
    uint32_t* masks = mask_bytes <= kSmallMaskBytes ? small_mask_scratch(s, mask_bytes) : nullptr;
    const bool pooled = masks == nullptr;
    if (pooled && (hipMallocAsync(reinterpret_cast<void**>(&masks), mask_bytes, s) != hipSuccess || !masks)) {
        (void)hipGetLastError();
        set_error("hash_encode_backward: no memory for %zu bytes of range masks", mask_bytes);
        return ENVIDR_ELAUNCH;
    }
    hipLaunchKernelGGL((k_table_range_masks<D>), dim3(ceil_div(B, kBlock), L), dim3(kBlock), 0, s, inputs, offsets, masks, B, pitch, log2_rows, ls);
    int rc = check_launch("k_table_range_masks");
    if (rc) { if (pooled) (void)hipFreeAsync(masks, s); return rc; }
    hipLaunchKernelGGL((k_table_scatter_lds<D, C, SECOND>), dim3(kXcds * kLdsScatterSlots), dim3(kLdsScatterThreads), 0, s, grad, inputs, offsets, ggx,
                       masks, pitch, grad_table, B, L, ls);
    rc = check_launch("k_table_scatter_lds");
    if (pooled) (void)hipFreeAsync(masks, s);             // stream order: behind the scatter kernel
    return rc;
}

// Per-point atomics run at ~33 G/s however the levels are spread (the fine levels' random rows: ~21 G/s each, one after the other --
// tools/probe/scatter_levels_probe.py); the LDS ranges cost ~0.2 ms (masks, one pass over the table) plus a quarter of that per point.  The
// two meet at ~35 k points (profiles/r05l/scatter_threshold_probe.txt); a training batch (146 k samples) is 0.89 -> 0.53 ms.
constexpr uint32_t kLdsScatterMinPoints = 1u << 15;

// ------------------------------------------------------------------------------------------
// dispatch helpers
// ------------------------------------------------------------------------------------------
template <typename F>
int dispatch_dc(uint32_t D, uint32_t C, const char* who, F&& f) {
#define ENVIDR_CASE(DD, CC) \
    if (D == DD && C == CC) return f(std::integral_constant<int, DD>{}, std::integral_constant<int, CC>{});
    ENVIDR_CASE(2, 1) ENVIDR_CASE(2, 2) ENVIDR_CASE(2, 4) ENVIDR_CASE(2, 8)
    ENVIDR_CASE(3, 1) ENVIDR_CASE(3, 2) ENVIDR_CASE(3, 4) ENVIDR_CASE(3, 8)
#undef ENVIDR_CASE
    set_error("%s: unsupported (D=%u, C=%u); D must be 2 or 3 and C one of 1, 2, 4, 8", who, D, C);
    return ENVIDR_EINVAL;
}

}  // namespace

extern "C" {

// Hands back what the library keeps between calls outside the caller's allocator (the range-mask scratch of the table-gradient scatters:
// one buffer per (device, stream), as large as the largest batch seen, at most 64 MiB).  Waits for the device.  Returns the bytes released.
uint64_t envidr_release_scratch(void) { return (uint64_t)release_small_mask_scratch(); }


int envidr_hash_encode_forward(const float* inputs, const float* embeddings, const int32_t* offsets, float* outputs,
                               uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                               int calc_grad_inputs, float* dy_dx, envidr_stream_t stream) {
    ENVIDR_REQUIRE(L >= 1 && L <= (uint32_t)kMaxLevels, "hash_encode_forward: L=%u out of range [1,%d]", L, kMaxLevels);
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(inputs && embeddings && offsets && outputs, "hash_encode_forward: null pointer");
    ENVIDR_REQUIRE(!calc_grad_inputs || dy_dx, "hash_encode_forward: dy_dx is null but calc_grad_inputs is set");
    const LevelScale ls = make_level_scale(L, S, H);
    const uint32_t chunks = ceil_div(B, kBlock);
    const dim3 grid(L * chunks);
    return dispatch_dc(D, C, "hash_encode_forward", [&](auto d, auto c) {
        constexpr int DD = decltype(d)::value, CC = decltype(c)::value;
        const uint32_t row_floats = L * DD * CC;
        constexpr bool kRowsAlwaysFit = kMaxLevels * DD * CC <= 127;     // (then the level-per-workgroup form with dy_dx is not instantiated at all)
        if (calc_grad_inputs && (kRowsAlwaysFit || row_floats <= 127)) { // the rows of 128 points fit the LDS budget (64 KiB): full-line dy_dx stores
            hipLaunchKernelGGL((k_hash_forward_rows<DD, CC>), dim3(ceil_div(B, kRowsPoints)), dim3(kBlock), kRowsPoints * (row_floats | 1u) * sizeof(float),
                               as_stream(stream), inputs, embeddings, offsets, outputs, B, L, ls, dy_dx, row_division_magic(row_floats));
        } else if (calc_grad_inputs) {
            if constexpr (!kRowsAlwaysFit)
                hipLaunchKernelGGL((k_hash_forward<DD, CC, true>), grid, dim3(kBlock), 0, as_stream(stream), inputs,
                                   embeddings, offsets, outputs, B, L, ls, chunks, dy_dx);
        } else
            hipLaunchKernelGGL((k_hash_forward<DD, CC, false>), grid, dim3(kBlock), 0, as_stream(stream), inputs,
                               embeddings, offsets, outputs, B, L, ls, chunks, dy_dx);
        return check_launch("k_hash_forward");
    });
}

int envidr_hash_encode_backward(const float* grad, const float* inputs, const float* embeddings,
                                const int32_t* offsets, float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                uint32_t L, float S, uint32_t H, int calc_grad_inputs, const float* dy_dx,
                                float* grad_inputs, envidr_stream_t stream) {
    (void)embeddings;
    ENVIDR_REQUIRE(L >= 1 && L <= (uint32_t)kMaxLevels, "hash_encode_backward: L=%u out of range", L);
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(grad && inputs && offsets, "hash_encode_backward: null pointer");
    ENVIDR_REQUIRE(!calc_grad_inputs || (dy_dx && grad_inputs), "hash_encode_backward: dy_dx/grad_inputs null");
    const LevelScale ls = make_level_scale(L, S, H);
    const uint32_t chunks = ceil_div(B, kBlock);
    return dispatch_dc(D, C, "hash_encode_backward", [&](auto d, auto c) {
        constexpr int DD = decltype(d)::value, CC = decltype(c)::value;
        if (grad_embeddings && B >= kLdsScatterMinPoints) {
            const int rc = launch_table_scatter_lds<DD, CC, false>(grad, inputs, offsets, nullptr, grad_embeddings, B, L, ls, as_stream(stream));
            if (rc) return rc;
        } else if (grad_embeddings) {
            hipLaunchKernelGGL((k_hash_backward_table<DD, CC>), dim3(xcd_grid_blocks(L, chunks)), dim3(kBlock), 0,
                               as_stream(stream), grad, inputs, offsets, grad_embeddings, B, L, ls, chunks);
            const int rc = check_launch("k_hash_backward_table");
            if (rc) return rc;
        }
        if (calc_grad_inputs) {
            launch_input_backward<DD, CC>(grad, dy_dx, grad_inputs, B, L, as_stream(stream));
            return check_launch("k_input_backward");
        }
        return ENVIDR_OK;
    });
}

int envidr_hash_encode_second_backward(const float* grad, const float* inputs, const float* embeddings,
                                       const int32_t* offsets, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                       float S, uint32_t H, int calc_grad_inputs, const float* dy_dx,
                                       const float* grad_grad_inputs, float* grad_grad, float* grad2_embeddings,
                                       envidr_stream_t stream) {
    (void)embeddings; (void)calc_grad_inputs;
    ENVIDR_REQUIRE(L >= 1 && L <= (uint32_t)kMaxLevels, "hash_encode_second_backward: L=%u out of range", L);
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(grad && inputs && offsets && dy_dx && grad_grad_inputs && grad_grad && grad2_embeddings,
                   "hash_encode_second_backward: null pointer");
    ENVIDR_REQUIRE(C != 1, "hash_encode_second_backward: C=1 is not supported (reference: hashencoder.cu:673-679)");
    const LevelScale ls = make_level_scale(L, S, H);
    const uint32_t chunks = ceil_div(B, kBlock);
    return dispatch_dc(D, C, "hash_encode_second_backward", [&](auto d, auto c) {
        constexpr int DD = decltype(d)::value, CC = decltype(c)::value;
        if constexpr (CC == 1) return (int)ENVIDR_EINVAL;         // refused above: no kernel is instantiated for the case
        else {
            hipLaunchKernelGGL((k_second_backward_grad<DD, CC>), dim3(chunks, L), dim3(kBlock), 0, as_stream(stream),
                               grad_grad_inputs, dy_dx, grad_grad, B, L);
            int rc = check_launch("k_second_backward_grad");
            if (rc) return rc;
            if (B >= kLdsScatterMinPoints) {
                return launch_table_scatter_lds<DD, CC, true>(grad, inputs, offsets, grad_grad_inputs, grad2_embeddings, B, L, ls, as_stream(stream));
            }
            hipLaunchKernelGGL((k_second_backward_table<DD, CC>), dim3(xcd_grid_blocks(L, chunks)), dim3(kBlock), 0,
                               as_stream(stream), grad, inputs, offsets, grad_grad_inputs, grad2_embeddings, B, L, ls,
                               chunks);
            return check_launch("k_second_backward_table");
        }
    });
}

}  // extern "C"
