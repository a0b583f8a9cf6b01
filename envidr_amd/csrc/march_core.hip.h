// Occupancy-grid ray marcher core, shared by the standalone operators (raymarching.hip) and the
// fused persistent render kernel (fused_render.hip).
//
// Semantics follow the reference's inference / training marchers
// (raymarching/src/raymarching.cu:839-944 and :340-508): per step clamp the position to the
// scene cube, pick the cascade level, truncate to a voxel, test one bit of the Morton-ordered
// bitfield, then either emit a sample and advance by dt or hop to the voxel's exit face.
//
// Bit-exactness contract (tests/test_ops_gpu.py): every float op below is a single IEEE
// fp32 (or, where the reference's expression promotes, fp64) operation in the reference's order;
// this TU is built with -ffp-contract=off and HIP's correctly rounded fp32 division.
#pragma once
#include "common.hip.h"

namespace envidr {

struct MarchConsts {
    float bound;
    float dt_gamma;
    float dt_min;      // 2*sqrt(3)/max_steps
    float dt_max;      // 2*sqrt(3)*2^(C-1)/H
    float cascades;    // C as float (the reference passes it through float parameters)
    float Hf;          // H as float
    float rH;          // 1/H
    float H3;          // H*H*H evaluated in uint32 then converted, like the reference's `H * H * H`
    float half_H;      // 0.5 * H
    float rbound;      // 1 / bound (correctly rounded, like the division it replaces)
    uint32_t H;
    uint32_t H_pow2;   // H is a power of two (the reference's grid_size is always 128; Morton indexing of the bitfield does
                       // not even stay inside H^3 otherwise, so the fp64 path below is kept only for arithmetic fidelity)
    const uint8_t* __restrict__ grid;
};

__host__ __device__ inline MarchConsts make_march_consts(float bound, float dt_gamma, uint32_t max_steps,
                                                        uint32_t C, uint32_t H, const uint8_t* grid) {
    MarchConsts k;
    const float two_sqrt3 = 2 * 1.7320508075688772f;
    k.bound = bound;
    k.dt_gamma = dt_gamma;
    k.dt_min = two_sqrt3 / max_steps;
    k.dt_max = two_sqrt3 * (1 << (C - 1)) / H;
    k.cascades = (float)C;
    k.Hf = (float)H;
    k.rH = 1 / (float)H;
    k.H3 = (float)(H * H * H);
    k.half_H = 0.5f * (float)H;
    k.rbound = 1 / bound;
    k.H = H;
    k.H_pow2 = (H & (H - 1)) == 0 ? 1u : 0u;
    k.grid = grid;
    return k;
}

struct RayGeom {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
};

__device__ __forceinline__ RayGeom load_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                            uint32_t id) {
    RayGeom r;
    const float* o = rays_o + (size_t)id * 3;
    const float* d = rays_d + (size_t)id * 3;
    r.ox = o[0]; r.oy = o[1]; r.oz = o[2];
    r.dx = d[0]; r.dy = d[1]; r.dz = d[2];
    r.rdx = 1 / r.dx; r.rdy = 1 / r.dy; r.rdz = 1 / r.dz;
    return r;
}

__device__ __forceinline__ float step_size(const MarchConsts& k, float t) {
    return clampf(t * k.dt_gamma, k.dt_min, k.dt_max);
}

// frexp exponent clamped to [0, C-1]; the reference routes this through float min/max.
__device__ __forceinline__ int clamp_level(const MarchConsts& k, float mag) {
    int e;
    frexpf(mag, &e);
    return (int)fminf(k.cascades - 1, fmaxf(0, (float)e));
}

// voxel coordinate along one axis: the reference evaluates 0.5 * (p * rbound + 1) * H in double
// (the literal 0.5 promotes), narrows to float for the clamp, then truncates to int.
// POW2: H is a power of two, so 0.5 * inner * H is a power-of-two scaling of an fp32 value: exact in double, exactly
// representable in fp32, hence equal to the single fp32 multiplication inner * (H / 2) -- same bits, no fp64 on the path.
template <bool POW2>
__device__ __forceinline__ int voxel_coord(const MarchConsts& k, float p, float mip_rbound) {
    const float inner = p * mip_rbound + 1;
    const float scaled = POW2 ? inner * k.half_H : (float)(0.5 * (double)inner * (double)k.H);
    return (int)clampf(scaled, 0.0f, (float)(k.H - 1));
}

// distance (in t) to the exit face of voxel coordinate n along one axis
__device__ __forceinline__ float exit_time(const MarchConsts& k, int n, float d, float rd, float p, float mip_bound) {
    return (((n + 0.5f + 0.5f * copysignf(1.0f, d)) * k.rH * 2 - 1) * mip_bound - p) * rd;
}

// One VISIT of the marcher's loop -- the body of the reference's `while (t < far)` (raymarching.cu:885-940) -- for the cell
// the ray is in at time t (the caller has checked t < far).  Occupied: returns true with the clamped sample position and
// the step `dt`, t unchanged (the caller emits the sample and advances by dt).  Empty: returns false with t moved to where
// the loop lands next (past the voxel's exit face, in whole steps).  march_next_* below are loops over this; kernels whose
// lanes would otherwise wait for each other inside nested per-sample loops call it directly, one visit per lane per
// iteration (geometry_pass.hip).
template <bool POW2>
__device__ __forceinline__ bool march_visit(const MarchConsts& k, const RayGeom& r, float& t, float& x, float& y, float& z, float& dt) {
    x = clampf(r.ox + t * r.dx, -k.bound, k.bound);
    y = clampf(r.oy + t * r.dy, -k.bound, k.bound);
    z = clampf(r.oz + t * r.dz, -k.bound, k.bound);
    dt = step_size(k, t);

    const int lvl_pos = clamp_level(k, fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))));
    // the reference's (dt * H) * 0.5 promotes to double: halving is exact, so the fp32 product has the same bits
    const int lvl_dt = clamp_level(k, (dt * k.Hf) * 0.5f);
    const int level = max(lvl_pos, lvl_dt);

    // mip_bound = min(2^level, bound); its reciprocal is exactly 2^-level, or the precomputed 1 / bound: the reference's
    // `1 / mip_bound` without a division per step
    const float pow2 = scalbnf(1.0f, level);
    const float mip_bound = fminf(pow2, k.bound);
    const float mip_rbound = pow2 <= k.bound ? scalbnf(1.0f, -level) : k.rbound;

    const int nx = voxel_coord<POW2>(k, x, mip_rbound);
    const int ny = voxel_coord<POW2>(k, y, mip_rbound);
    const int nz = voxel_coord<POW2>(k, z, mip_rbound);

    // level * H3 + morton is a float expression in the reference (H3 is float)
    const uint32_t bit = (uint32_t)((float)level * k.H3 + (float)morton_encode(nx, ny, nz));
    if (k.grid[bit >> 3] & (1u << (bit & 7))) return true;

    const float tx = exit_time(k, nx, r.dx, r.rdx, x, mip_bound);
    const float ty = exit_time(k, ny, r.dy, r.rdy, y, mip_bound);
    const float tz = exit_time(k, nz, r.dz, r.rdz, z, mip_bound);
    const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    do {
        t += step_size(k, t);
    } while (t < tt);
    return false;
}

// Advance `t` to the next occupied sample strictly before `far`.
// On success: (x,y,z) is the clamped sample position, dt the step used for alpha, and `t` has
// already been advanced past the sample (t_after = t_sample + dt).  Returns false when the ray
// leaves [.., far) without another occupied sample.
// `t_at` (optional) receives the ray time AT the emitted sample (before the += dt): resuming the
// marcher from exactly that value re-emits the same sample, which is how the fused renderer's
// first-hit pre-pass hands rays over without changing any arithmetic.
template <bool POW2>
__device__ __forceinline__ bool march_next_impl(const MarchConsts& k, const RayGeom& r, float far, float& t,
                                                float& x, float& y, float& z, float& dt, float* t_at) {
    while (t < far) {
        if (march_visit<POW2>(k, r, t, x, y, z, dt)) {
            if (t_at) *t_at = t;
            t += dt;
            return true;
        }
    }
    return false;
}

// one wave-uniform choice per call, outside the marching loops
__device__ __forceinline__ bool march_next(const MarchConsts& k, const RayGeom& r, float far, float& t,
                                           float& x, float& y, float& z, float& dt, float* t_at = nullptr) {
    return k.H_pow2 ? march_next_impl<true>(k, r, far, t, x, y, z, dt, t_at) : march_next_impl<false>(k, r, far, t, x, y, z, dt, t_at);
}

// The same marcher for the geometry every scene of the reference has -- ONE cascade (bound <= 1: every level clamp gives 0,
// mip_bound = bound, 1 / mip_bound = 1 / bound) on a power-of-two grid of at most 256^3 cells -- reading a copy of the
// bitfield in x-fastest LINEAR cell order (k_linearize_bitfield) instead of Morton order.  Every float operation that
// produces t, the position, dt or the voxel coordinates is the statement of march_visit<true>, so the samples are the
// same bits; what is gone is the arithmetic whose result is a constant here (two frexp level clamps, the scalbn pair, the
// float bit index) and the 3 x 10-operation bit interleave.  These kernels are VALU-issue bound (a ray that misses the
// object walks ~200 empty cells), so the instruction count per cell is their run time.
// GAMMA0: dt_gamma == 0, the step is one constant (clampf(t * 0, dt_min, dt_max) for any finite t).
__device__ __forceinline__ bool march_fast_ok(const MarchConsts& k) { return k.cascades == 1.0f && k.H_pow2 && k.H <= 256u; }

template <bool GAMMA0>
__device__ __forceinline__ bool march_visit_c1(const MarchConsts& k, const uint8_t* __restrict__ linear_grid, uint32_t log2H, const RayGeom& r,
                                               float& t, float& x, float& y, float& z, float& dt) {
    const float dtc = step_size(k, 0.0f);
    x = clampf(r.ox + t * r.dx, -k.bound, k.bound);
    y = clampf(r.oy + t * r.dy, -k.bound, k.bound);
    z = clampf(r.oz + t * r.dz, -k.bound, k.bound);
    dt = GAMMA0 ? dtc : step_size(k, t);
    const int nx = voxel_coord<true>(k, x, k.rbound);
    const int ny = voxel_coord<true>(k, y, k.rbound);
    const int nz = voxel_coord<true>(k, z, k.rbound);
    const uint32_t bit = (uint32_t)nx | ((uint32_t)ny << log2H) | ((uint32_t)nz << (2 * log2H));
    if (linear_grid[bit >> 3] & (1u << (bit & 7))) return true;
    const float tx = exit_time(k, nx, r.dx, r.rdx, x, k.bound);
    const float ty = exit_time(k, ny, r.dy, r.rdy, y, k.bound);
    const float tz = exit_time(k, nz, r.dz, r.rdz, z, k.bound);
    const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    if constexpr (GAMMA0) {
        do { t += dtc; } while (t < tt);
    } else {
        do { t += step_size(k, t); } while (t < tt);
    }
    return false;
}

template <bool GAMMA0>
__device__ __forceinline__ bool march_next_c1(const MarchConsts& k, const uint8_t* __restrict__ linear_grid, uint32_t log2H, const RayGeom& r,
                                              float far, float& t, float& x, float& y, float& z, float& dt, float* t_at) {
    while (t < far) {
        if (march_visit_c1<GAMMA0>(k, linear_grid, log2H, r, t, x, y, z, dt)) {
            if (t_at) *t_at = t;
            t += dt;
            return true;
        }
    }
    return false;
}

// One compositing update (reference raymarching.cu:996-1030, inference form: transmittance is
// re-derived from the running weight sum, termination is tested with the pre-update T).
struct Accum {
    float ws, depth, r, g, b, t;
};

__device__ __forceinline__ float alpha_from_sigma(float sigma, float delta, uint32_t input_alpha) {
    return input_alpha ? 0.0f + sigma : 1.0f - __expf(-sigma * delta);
}

// returns true when the ray must terminate AFTER this sample (T < T_thresh)
__device__ __forceinline__ bool composite_sample(Accum& a, float alpha, float delta_depth, float cr, float cg,
                                                 float cb, float T_thresh, uint32_t accum_deltas) {
    const float T = 1 - a.ws;
    const float w = alpha * T;
    a.ws += w;
    a.t = accum_deltas ? a.t + delta_depth : delta_depth;
    a.depth += w * a.t;
    a.r += w * cr;
    a.g += w * cg;
    a.b += w * cb;
    return T < T_thresh;
}

}  // namespace envidr
