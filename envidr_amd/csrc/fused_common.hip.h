// Per-sample device code shared by the fused render kernels (fused_render.hip) and the geometry pipeline
// (geometry_pass.hip): slab test, hash-grid level evaluation split into "issue the gathers" / "interpolate",
// the fragment layout of the SDF weight blob, small math helpers.  Both translation units evaluate a sample with
// exactly these statements, so their per-sample results are the same bits.
#pragma once
#include "grid_core.hip.h"
#include "march_core.hip.h"
#include "mlp_mfma.hip.h"
#include "env_pass.hip.h"

#include "../../include/envidr_render.h"

#include <float.h>

namespace envidr {

constexpr int kLevels = ENVIDR_MAX_LEVELS;

struct HashLevelK {
    uint32_t row0, size, stride1, stride2;
    float scale;
    uint32_t hashed, andmask, enabled, slow_mod;
};

// slab test, identical arithmetic to k_near_far_from_aabb (raymarching.hip; reference raymarching.cu:91-145).
// aabb = {xmin, ymin, zmin, xmax, ymax, zmax}: the model's aabb_infer, which a tightened marching box or a checkpoint may
// make smaller than [-bound, bound]^3 (the marcher itself still clamps positions to the cube of half extent `bound`).
struct Aabb { float lo[3], hi[3]; };
__host__ __device__ inline Aabb make_aabb(const envidr_render_desc* d) {
    Aabb b;
    for (int i = 0; i < 3; ++i) { b.lo[i] = d->has_aabb ? d->aabb[i] : -d->bound; b.hi[i] = d->has_aabb ? d->aabb[3 + i] : d->bound; }
    return b;
}
__device__ __forceinline__ void near_far(const RayGeom& r, const Aabb& box, float min_near, float& near, float& far) {
    near = (box.lo[0] - r.ox) * r.rdx; far = (box.hi[0] - r.ox) * r.rdx;
    if (near > far) { const float c = near; near = far; far = c; }
    float ny = (box.lo[1] - r.oy) * r.rdy, fy = (box.hi[1] - r.oy) * r.rdy;
    if (ny > fy) { const float c = ny; ny = fy; fy = c; }
    bool miss = near > fy || ny > far;
    if (!miss) {
        if (ny > near) near = ny;
        if (fy < far) far = fy;
        float nz = (box.lo[2] - r.oz) * r.rdz, fz = (box.hi[2] - r.oz) * r.rdz;
        if (nz > fz) { const float c = nz; nz = fz; fz = c; }
        miss = near > fz || nz > far;
        if (!miss) {
            if (nz > near) near = nz;
            if (fz < far) far = fz;
            if (near < min_near) near = min_near;
        }
    }
    if (miss) near = far = FLT_MAX;
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float softplusf(float x) { return x > 20.0f ? x : log1pf(expf(x)); }   // torch: beta 1, threshold 20

template <int N>
__device__ __forceinline__ void normalize_n(float (&v)[N], float eps) {
    float s = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) s += v[i] * v[i];
    const float inv = 1.0f / fmaxf(sqrtf(s), eps);    // F.normalize: v / max(||v||, eps)
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] * inv;
}

// position of the n-th set bit of mask (n < popcount(mask))
__device__ __forceinline__ uint32_t nth_set_bit(unsigned long long mask, uint32_t n) {
    uint32_t pos = 0;
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
        const uint32_t cnt = (uint32_t)__popcll((mask >> pos) & ((1ull << w) - 1ull));
        if (n >= cnt) { n -= cnt; pos += (uint32_t)w; }
    }
    return pos;
}

// ---- hash-grid level evaluation split into "issue the gathers" and "interpolate" ------------------
struct HashStage {
    float w1[3], dw[3];
    float2 c[8];
};

// Row index -> row inside the level's table, branch-free for the two geometries real tables have: hashed levels have
// a power-of-two size (mask), dense levels produce indices below 2 * size (host-checked), where one conditional
// subtract -- written as min(idx, idx - size) on unsigned values -- is the modulo.  `andmask` is size - 1 or ~0.
// (Run-time branches per corner split this section into dozens of basic blocks; the waits the compiler then places at
// their joins drained the gather pipeline.)
__device__ __forceinline__ uint32_t wrap_fast(uint32_t idx, const HashLevelK& lv) {
    idx &= lv.andmask;
    return min(idx, idx - lv.size);
}

template <bool SLOW>
__device__ __forceinline__ void hash_gather(const HashLevelK& lv, const float2* __restrict__ table, const uint32_t (&cell)[3], HashStage& st) {
    if (lv.hashed) {
        const uint32_t hx[2] = {cell[0], cell[0] + 1u};
        const uint32_t hy[2] = {cell[1] * 2654435761u, (cell[1] + 1u) * 2654435761u};
        const uint32_t hz[2] = {cell[2] * 805459861u, (cell[2] + 1u) * 805459861u};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t idx = hx[i & 1] ^ hy[(i >> 1) & 1] ^ hz[(i >> 2) & 1];
            st.c[i] = table[SLOW ? idx % lv.size : wrap_fast(idx, lv)];
        }
    } else {
        const uint32_t ix[2] = {cell[0], cell[0] + 1u};
        const uint32_t iy[2] = {cell[1] * lv.stride1, (cell[1] + 1u) * lv.stride1};
        const uint32_t iz[2] = {cell[2] * lv.stride2, (cell[2] + 1u) * lv.stride2};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t idx = ix[i & 1] + iy[(i >> 1) & 1] + iz[(i >> 2) & 1];
            st.c[i] = table[SLOW ? idx % lv.size : wrap_fast(idx, lv)];
        }
    }
}

// `table_base` is the whole table ([rows, 2] floats); x is the position in [0, 1]^3
__device__ __forceinline__ void hash_prepare(const HashLevelK& lv, const float* __restrict__ table_base, const float (&x)[3], HashStage& st) {
    uint32_t cell[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float p = x[d] * lv.scale + 0.0f;
        cell[d] = (uint32_t)floorf(p);
        p -= (float)cell[d];
        st.dw[d] = 6 * p * (1.0f - p);                 // smoothstep'
        st.w1[d] = p * p * (3.0f - 2.0f * p);          // smoothstep
    }
    const float2* table = reinterpret_cast<const float2*>(table_base) + lv.row0;
    if (lv.slow_mod) hash_gather<true>(lv, table, cell, st);       // table geometries that are neither (never for HashEncoder's own sizing)
    else hash_gather<false>(lv, table, cell, st);
}

__device__ __forceinline__ void hash_finish(const float scale, const HashStage& st, float (&out)[2], float (&dydx)[3][2]) {
    out[0] = out[1] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float w = 1;
#pragma unroll
        for (int d = 0; d < 3; ++d) w *= ((i >> d) & 1) ? st.w1[d] : 1 - st.w1[d];
        out[0] += w * st.c[i].x;
        out[1] += w * st.c[i].y;
    }
#pragma unroll
    for (int gd = 0; gd < 3; ++gd) {
        float acc0 = 0, acc1 = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float w = scale;
            int lo = 0;
#pragma unroll
            for (int nd = 0; nd < 2; ++nd) {
                const int d = nd >= gd ? nd + 1 : nd;
                const int bit = (j >> nd) & 1;
                w *= bit ? st.w1[d] : 1 - st.w1[d];
                lo |= bit << d;
            }
            const int hi = lo | (1 << gd);
            acc0 += w * (st.c[hi].x - st.c[lo].x) * st.dw[gd];
            acc1 += w * (st.c[hi].y - st.c[lo].y) * st.dw[gd];
        }
        dydx[gd][0] = acc0;
        dydx[gd][1] = acc1;
    }
}

// host: per-level constants of the fused kernels from the render descriptor; returns an error message or nullptr
inline const char* fill_hash_levels(const envidr_render_desc* d, HashLevelK (&lv)[kLevels]) {
    const LevelScale ls = make_level_scale(d->num_levels, d->log2_per_level_scale, d->base_resolution);
    for (uint32_t l = 0; l < d->num_levels; ++l) {
        const uint32_t size = (uint32_t)(d->hash_offsets[l + 1] - d->hash_offsets[l]);
        const LevelGeom<3> g = make_level_geom<3>(size, ls.resolution[l], true);
        lv[l].row0 = (uint32_t)d->hash_offsets[l];
        lv[l].size = size;
        lv[l].stride1 = g.stride[1]; lv[l].stride2 = g.stride[2];
        lv[l].scale = ls.scale[l];
        lv[l].hashed = g.hashed;
        lv[l].andmask = (g.hashed && g.pow2) ? size - 1u : 0xffffffffu;
        {
            // dense levels: the largest index a corner can produce (coordinate res on every axis) must stay below 2 * size
            // for the conditional-subtract wrap; otherwise fall back to a true modulo
            const unsigned long long res = ls.resolution[l];
            const unsigned long long max_idx = res + res * (unsigned long long)g.stride[1] + res * (unsigned long long)g.stride[2];
            lv[l].slow_mod = ((g.hashed && !g.pow2) || (!g.hashed && max_idx >= 2ull * size)) ? 1u : 0u;
        }
        lv[l].enabled = (d->enabled_levels <= 0 || (int32_t)l < d->enabled_levels) ? 1u : 0u;
        if (!(g.hashed || g.stride[0] == 1)) return "unexpected dense stride";
    }
    return nullptr;
}

// fragment layout of the SDF weight pass (must match envidr_amd/fused.py and envidr_render.h)
// (every forward layer carries its bias as one extra leading step; the two gradient layers have none)
constexpr int kSdfW1 = 0, kSdfW2 = kSdfW1 + lane_layer_frags(16, 2, true), kSdfW3 = kSdfW2 + tile_layer_frags(2, 2, true),
              kSdfW2t = kSdfW3 + tile_layer_frags(2, 1, true), kSdfW1t = kSdfW2t + tile_layer_frags(2, 2, false),
              kSdfFrags = kSdfW1t + tile_layer_frags(2, 1, false);

// arguments of the shade-only kernels (fused_render.hip k_shade_samples, shade_split.hip k_env_split)
struct ShadeArgs {
    const float* normals;       // [M,3] unit
    const float* dirs;          // [M,3] unit view directions (camera -> sample)
    const float* geo_feat;      // [M,12] (stride 12) or one shared [12] (stride 0), already unit-normalised
    const float* roughness;     // [M] (stride 1) or one shared value (stride 0): the IDE kappa_inv of the reflected direction
    uint32_t geo_stride, rough_stride, M;
    // record mode (two-phase frames): the view direction of record i is rays_d[ray_ids[i]] and the number of records is
    // read on the device (min(*m_dev, M)), so the host never waits for the geometry pass
    const uint32_t* ray_ids;
    const float* rays_d;
    const uint32_t* m_dev;
    const uint32_t* slot;       // record mode, optional: normals / geo_feat / roughness of record i live at index slot[i]
    const uint32_t* list;       // record mode, optional: the records to shade (the i-th shaded record is list[i]; m_dev counts the list)
    uint32_t* work;             // device word, zero at launch: the next unclaimed record (waves claim 64 at a time; null: static grid stride)
    const uint32_t* m_all;      // with list: the number of records; when the list holds them all it is the identity and is not read
    // reflected-radiance branch (record mode): per-ray (rgb, visibility), the learnt blend logit per sample, the two extra blobs
    const float* r_images; const float* blend; const float* renv_blob; const float* spec2_blob;
    float rough_scale, indir_rough_thresh;
    const float* env_blob;
    const float* head_blob;
    float kappa_diffuse, light_scale;
    int has_rot;
    float rot[9];
    float* c_diffuse;           // [M,3]
    float* c_specular;          // [M,3]
    // split-precision mode (shade_split.hip): the environment features of sample i, [M,24] = env(normal) | env(reflection),
    // written by k_env_split and consumed by the PRE_ENV instantiation of k_shade_samples
    float* env_pre;
};

int device_cu_count();   // fused_render.hip
int launch_env_split(const envidr_render_desc* d, const ShadeArgs& a, hipStream_t s, const char* who);   // shade_split.hip
int launch_env_split2(const envidr_render_desc* d, const ShadeArgs& a, hipStream_t s, const char* who);  // shade_split2.hip

}  // namespace envidr
