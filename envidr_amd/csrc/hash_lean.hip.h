// Lean evaluation of one smoothstep hash-grid level (value + analytic Jacobian) for the sample-parallel geometry
// kernel (geometry_pass.hip).  Same function as hashencoder/src/hashencoder.cu:103-254 (kernel_grid, D = 3, C = 2) --
// same cell, same smoothstep weights, same eight corner rows -- organised for the vector ALU budget of a kernel that
// shares its SIMDs with the matrix pipe:
//   * corner rows are gathered through ONE buffer descriptor over the whole table (one 8-byte load per row): 32-bit byte
//     offsets in one VGPR each, the level's first row as the scalar offset; no 64-bit address arithmetic;
//   * the only data-dependent control flow is one wave-uniform choice (dense or hashed level) around eight integer
//     index computations; hashed levels must have a power-of-two size and dense levels must satisfy the
//     conditional-subtract wrap (HashEncoder's own sizing always does; the host checks and refuses otherwise);
//   * value and gradient come from one factorised trilinear pass (lerp along x, then y, then z; the differences the
//     lerps need ARE the partial derivatives): ~60 vector-ALU operations per level instead of ~250 for the expanded
//     sum over corners.  The result differs from the expanded sum by fp32 rounding only (tests/test_geometry_gpu.py).
#pragma once
#include "fused_common.hip.h"

namespace envidr {

struct LeanLevel {
    uint32_t row0_bytes;    // first row of the level, in bytes from the table base
    uint32_t size;          // rows
    uint32_t m1, m2;        // dense: strides of y and z; hashed: unused
    float scale;
    uint32_t mask;          // hashed: size - 1
    uint32_t hashed;
    float on;               // 1 = level enabled, 0 = masked out (network.py:390-393)
};

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// One level in flight: weights and the raw gathers, one 8-byte load per corner row.  (Round 3 fetched the two x-neighbours of
// a (y, z) corner pair with ONE 16-byte load whenever their rows were adjacent -- 6 line lookups per level instead of 8 -- and
// picked the halves apart with ~45 selects and mask operations per level; since the geometry kernel became issue-bound the
// selects cost more than the two gathers: 2.84 -> 2.61 ms per headline frame without them, tools/geo/build_variants.py history.)
struct LeanStage {
    float w[3];             // smoothstep weight of the +1 corner per axis
    float sdw[3];           // scale * d(w)/d(frac) per axis
    u32x2 row[8];           // corner rows, index bx | by << 1 | bz << 2
};

// host: returns an error text when the table geometry is outside what the lean path handles
inline const char* fill_lean_levels(const envidr_render_desc* d, LeanLevel (&out)[kLevels]) {
    HashLevelK lv[kLevels];
    memset(lv, 0, sizeof(lv));
    const char* err = fill_hash_levels(d, lv);
    if (err) return err;
    for (uint32_t l = 0; l < d->num_levels; ++l) {
        if (lv[l].size < 2) return "hash level with fewer than two rows";
        if (lv[l].slow_mod) return "hash level with neither a power-of-two hashed size nor a dense index range below 2 * size";
        out[l].row0_bytes = lv[l].row0 * 8u;
        out[l].size = lv[l].size;
        out[l].m1 = lv[l].stride1; out[l].m2 = lv[l].stride2;
        out[l].scale = lv[l].scale;
        out[l].mask = lv[l].size - 1u;
        out[l].hashed = lv[l].hashed;
        out[l].on = lv[l].enabled ? 1.0f : 0.0f;
        if ((unsigned long long)lv[l].row0 * 8ull + (unsigned long long)lv[l].size * 8ull > 0xffffffffull) return "hash table larger than 4 GiB";
    }
    return nullptr;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t table_rsrc(const float* table, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(table), 0, (int)bytes, 0x00020000);
}

// cell + weights + the gathers of one level (x in [0, 1]^3)
// LANE_LEVEL: the level differs between lanes (its constants live in vector registers): the level's first row then goes
// into the per-lane offset instead of the scalar offset of the buffer instruction
template <int AUX = 0, bool LANE_LEVEL = false>
// rows_out (tests only, null in every production call): the eight corner rows within the level, index bx | by << 1 | bz << 2
__device__ __forceinline__ void lean_prepare(const LeanLevel& lv, __amdgpu_buffer_rsrc_t table, const float (&x)[3], LeanStage& st,
                                             uint32_t* rows_out = nullptr) {
    uint32_t cell[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float p = x[d] * lv.scale + 0.0f;
        cell[d] = (uint32_t)floorf(p);
        p -= (float)cell[d];
        st.sdw[d] = (6 * p * (1.0f - p)) * lv.scale;     // smoothstep' * scale
        st.w[d] = p * p * (3.0f - 2.0f * p);             // smoothstep
    }
    uint32_t r0[4], r1[4];          // rows of the x and x + 1 corners of (y, z) pair j = by | bz << 1
    if (lv.hashed) {
        const uint32_t hy0 = cell[1] * 2654435761u, hy1 = hy0 + 2654435761u;
        const uint32_t hz0 = cell[2] * 805459861u, hz1 = hz0 + 805459861u;
        const uint32_t yz[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
        const uint32_t x0 = cell[0], x1 = cell[0] + 1u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r0[j] = (x0 ^ yz[j]) & lv.mask;
            r1[j] = (x1 ^ yz[j]) & lv.mask;
        }
    } else {
        const uint32_t y0 = cell[1] * lv.m1, y1 = y0 + lv.m1;
        const uint32_t z0 = cell[2] * lv.m2, z1 = z0 + lv.m2;
        const uint32_t yz[4] = {y0 + z0, y1 + z0, y0 + z1, y1 + z1};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t i0 = cell[0] + yz[j], i1 = i0 + 1u;
            r0[j] = min(i0, i0 - lv.size);                // idx mod size for idx < 2 size (unsigned wrap makes the minimum pick it)
            r1[j] = min(i1, i1 - lv.size);
        }
    }
    if (rows_out) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { rows_out[2 * j] = r0[j]; rows_out[2 * j + 1] = r1[j]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        st.row[2 * j] = __builtin_amdgcn_raw_buffer_load_b64(table, r0[j] * 8u + (LANE_LEVEL ? lv.row0_bytes : 0u), LANE_LEVEL ? 0u : lv.row0_bytes, AUX);
        st.row[2 * j + 1] = __builtin_amdgcn_raw_buffer_load_b64(table, r1[j] * 8u + (LANE_LEVEL ? lv.row0_bytes : 0u), LANE_LEVEL ? 0u : lv.row0_bytes, AUX);
    }
}

// the eight corner rows of a stage, index = bx | by << 1 | bz << 2
__device__ __forceinline__ void lean_corners(const LeanStage& st, float2 (&c)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { c[i].x = __uint_as_float(st.row[i][0]); c[i].y = __uint_as_float(st.row[i][1]); }
}

// value (2 channels) and d value / d x01 (3 x 2), multiplied by `m` (level mask x inside-the-cube mask).  The two channels of a
// row go through identical arithmetic: it is written on 2-vectors so that it compiles to packed fp32 instructions
// (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: two IEEE operations per issue slot -- the geometry kernel is issue-bound);
// per channel the operations and their order are those of the scalar form, so the bits are too.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ void lean_finish(const LeanStage& st, float m, f32x2& out, f32x2 (&g)[3]) {
    float2 cc[8];
    lean_corners(st, cc);
    f32x2 c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = f32x2{cc[i].x, cc[i].y};
    const f32x2 wx = {st.w[0], st.w[0]}, wy = {st.w[1], st.w[1]}, wz = {st.w[2], st.w[2]};
    const float sx = st.sdw[0] * m, sy = st.sdw[1] * m, sz = st.sdw[2] * m;
    f32x2 D[4], a[4];                 // x differences and x-interpolated values for the four (y, z) corners
#pragma unroll
    for (int j = 0; j < 4; ++j) { D[j] = c[2 * j + 1] - c[2 * j]; a[j] = pk_fma(wx, D[j], c[2 * j]); }
    f32x2 F[2], b[2], E[2];           // y differences / y-interpolated values / y-interpolated x differences for z = 0, 1
#pragma unroll
    for (int z = 0; z < 2; ++z) {
        F[z] = a[2 * z + 1] - a[2 * z];
        b[z] = pk_fma(wy, F[z], a[2 * z]);
        E[z] = pk_fma(wy, D[2 * z + 1] - D[2 * z], D[2 * z]);
    }
    const f32x2 G = b[1] - b[0];
    out = pk_fma(wz, G, b[0]) * f32x2{m, m};
    g[0] = pk_fma(wz, E[1] - E[0], E[0]) * f32x2{sx, sx};
    g[1] = pk_fma(wz, F[1] - F[0], F[0]) * f32x2{sy, sy};
    g[2] = G * f32x2{sz, sz};
}

}  // namespace envidr
