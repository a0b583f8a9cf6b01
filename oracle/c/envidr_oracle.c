/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement, for the CPU, of the reference algorithms on the ENVIDR render hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the checker / the timed CPU baseline -- never as (part of) the product path.
 *
 * Pinning: every function here is compared in tests/test_oracle_pinning.py against
 *   (a) oracle/_ref (the reference's own kernel bodies run on the CPU, built by
 *       oracle/ref/build_ref.py in the container that has /root/reference), and
 *   (b) the committed golden vectors under tests/golden/ generated from (a) and from the imported
 *       reference Python (tests/golden/make_golden.py).
 * The reference ships no tests or golden vectors of its own for this path.
 *
 * Entry points mirror include/envidr_amd.h (same argument order, host pointers, no stream), prefix
 * `oracle_`.  Loops over rays/points are OpenMP-parallel (the cpu_baseline reports the thread
 * count it used); arithmetic inside one ray/point is strictly sequential in the reference's order,
 * and this file is compiled with -ffp-contract=off so no multiply-add is fused.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))

static inline float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

/* ---- Morton code, 10 bits per axis (raymarching.cu:56-82) ---- */
static inline uint32_t spread3(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static inline uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) { return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2); }
static inline uint32_t compact3(uint32_t v) {
    v &= 0x49249249u;
    v = (v | (v >> 2)) & 0xC30C30C3u;
    v = (v | (v >> 4)) & 0x0F00F00Fu;
    v = (v | (v >> 8)) & 0xFF0000FFu;
    v = (v | (v >> 16)) & 0x0000FFFFu;
    return v;
}

/* ============================ raymarching ============================ */

/* raymarching.cu:91-145 */
API int oracle_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near,
                                  float* nears, float* fars) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; ++n) {
        const float* o = rays_o + 3 * n; const float* d = rays_d + 3 * n;
        const float rdx = 1 / d[0], rdy = 1 / d[1], rdz = 1 / d[2];
        float near = (aabb[0] - o[0]) * rdx, far = (aabb[3] - o[0]) * rdx, tmp;
        if (near > far) { tmp = near; near = far; far = tmp; }
        float ny = (aabb[1] - o[1]) * rdy, fy = (aabb[4] - o[1]) * rdy;
        if (ny > fy) { tmp = ny; ny = fy; fy = tmp; }
        if (near > fy || ny > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (ny > near) near = ny;
        if (fy < far) far = fy;
        float nz = (aabb[2] - o[2]) * rdz, fz = (aabb[5] - o[2]) * rdz;
        if (nz > fz) { tmp = nz; nz = fz; fz = tmp; }
        if (near > fz || nz > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (nz > near) near = nz;
        if (fz < far) far = fz;
        if (near < min_near) near = min_near;
        nears[n] = near; fars[n] = far;
    }
    return 0;
}

/* raymarching.cu:162-198 */
API int oracle_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords) {
    const float rpi = 0.3183098861837907f;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; ++n) {
        const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
        const float dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
        const float A = dx * dx + dy * dy + dz * dz;
        const float B = ox * dx + oy * dy + oz * dz;
        const float C = ox * ox + oy * oy + oz * oz - radius * radius;
        const float t = (-B + sqrtf(B * B - A * C)) / A;
        const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
        const float theta = (float)atan2((double)sqrtf(x * x + z * z), (double)y);
        const float phi = (float)atan2((double)z, (double)x);
        coords[2 * n] = 2 * theta * rpi - 1;
        coords[2 * n + 1] = phi * rpi;
    }
    return 0;
}

/* raymarching.cu:214-264 */
API int oracle_morton3D(const int32_t* coords, uint32_t N, int32_t* indices) {
    for (uint32_t n = 0; n < N; ++n) indices[n] = (int32_t)morton3((uint32_t)coords[3 * n], (uint32_t)coords[3 * n + 1], (uint32_t)coords[3 * n + 2]);
    return 0;
}
API int oracle_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords) {
    for (uint32_t n = 0; n < N; ++n) {
        const int32_t v = indices[n];
        coords[3 * n] = (int32_t)compact3((uint32_t)(v >> 0));
        coords[3 * n + 1] = (int32_t)compact3((uint32_t)(v >> 1));
        coords[3 * n + 2] = (int32_t)compact3((uint32_t)(v >> 2));
    }
    return 0;
}

/* raymarching.cu:267-289 */
API int oracle_packbits(const float* grid, uint32_t N, float thresh, uint8_t* bitfield) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; ++n) {
        uint8_t bits = 0;
        for (int i = 0; i < 8; ++i) bits |= (grid[8 * n + i] > thresh) ? (uint8_t)(1u << i) : 0;
        bitfield[n] = bits;
    }
    return 0;
}

/* raymarching.cu:302-321 */
API int oracle_get_scatter_idx(const int32_t* rays, uint32_t N, int32_t* idx_map) {
    for (uint32_t n = 0; n < N; ++n) {
        const uint32_t id = rays[3 * n], off = rays[3 * n + 1], cnt = rays[3 * n + 2];
        for (uint32_t s = 0; s < cnt; ++s) idx_map[off + s] = (int32_t)id;
    }
    return 0;
}

/* ---- the marcher's inner step, shared by the inference and training forms ---- */
typedef struct {
    float bound, dt_gamma, dt_min, dt_max, Cf, Hf, rH, H3;
    uint32_t H;
    const uint8_t* grid;
} march_k;

static march_k march_consts(float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid) {
    march_k k;
    const float two_sqrt3 = 2 * 1.7320508075688772f;           /* 2 * SQRT3()                       */
    k.bound = bound; k.dt_gamma = dt_gamma;
    k.dt_min = two_sqrt3 / max_steps;                           /* raymarching.cu:878                */
    k.dt_max = two_sqrt3 * (1 << (C - 1)) / H;                  /* raymarching.cu:879                */
    k.Cf = (float)C; k.Hf = (float)H; k.rH = 1 / (float)H;
    k.H3 = (float)(H * H * H);                                  /* `const float H3 = H * H * H;`     */
    k.H = H; k.grid = grid;
    return k;
}
static inline int level_of(const march_k* k, float mag) {       /* mip_from_pos / mip_from_dt :43-55 */
    int e;
    frexpf(mag, &e);
    return (int)fminf(k->Cf - 1, fmaxf(0, (float)e));
}
static inline int voxel_of(const march_k* k, float p, float rb) { /* :904-906, evaluated in double   */
    const float inner = p * rb + 1;
    return (int)clampf((float)(0.5 * (double)inner * (double)k->H), 0.0f, (float)(k->H - 1));
}
static inline float exit_of(const march_k* k, int n, float d, float rd, float p, float mb) { /* :933-935 */
    return (((n + 0.5f + 0.5f * copysignf(1.0f, d)) * k->rH * 2 - 1) * mb - p) * rd;
}
/* one pass of `while (t < far ...)` until an occupied sample is found; 1 = sample emitted */
static int march_next(const march_k* k, const float* o, const float* d, const float* rd, float far, float* t,
                      float* xyz, float* dt_out) {
    while (*t < far) {
        const float x = clampf(o[0] + *t * d[0], -k->bound, k->bound);
        const float y = clampf(o[1] + *t * d[1], -k->bound, k->bound);
        const float z = clampf(o[2] + *t * d[2], -k->bound, k->bound);
        const float dt = clampf(*t * k->dt_gamma, k->dt_min, k->dt_max);
        const int l0 = level_of(k, fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))));
        const int l1 = level_of(k, (float)((double)(dt * k->Hf) * 0.5));
        const int level = l0 > l1 ? l0 : l1;
        const float mb = fminf(scalbnf(1.0f, level), k->bound);
        const float rb = 1 / mb;
        const int nx = voxel_of(k, x, rb), ny = voxel_of(k, y, rb), nz = voxel_of(k, z, rb);
        const uint32_t bit = (uint32_t)((float)level * k->H3 + (float)morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
        if (k->grid[bit / 8] & (1 << (bit % 8))) {
            xyz[0] = x; xyz[1] = y; xyz[2] = z;
            *dt_out = dt;
            *t += dt;
            return 1;
        }
        const float tx = exit_of(k, nx, d[0], rd[0], x, mb);
        const float ty = exit_of(k, ny, d[1], rd[1], y, mb);
        const float tz = exit_of(k, nz, d[2], rd[2], z, mb);
        const float tt = *t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        do { *t += clampf(*t * k->dt_gamma, k->dt_min, k->dt_max); } while (*t < tt);
    }
    return 0;
}

/* raymarching.cu:839-944 */
API int oracle_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                          const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                          uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars, float* xyzs,
                          float* dirs, float* deltas, const float* noises) {
    (void)nears;
    const march_k k = march_consts(bound, dt_gamma, max_steps, C, H, grid);
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t n = 0; n < (int64_t)n_alive; ++n) {
        const uint32_t id = (uint32_t)rays_alive[n];
        const float* o = rays_o + 3 * (size_t)id; const float* d = rays_d + 3 * (size_t)id;
        const float rd[3] = {1 / d[0], 1 / d[1], 1 / d[2]};
        const float far = fars[id];
        float t = rays_t[id], last_t = t;
        t += clampf(t * k.dt_gamma, k.dt_min, k.dt_max) * noises[n];
        float* px = xyzs + (size_t)n * n_step * 3; float* pd = dirs + (size_t)n * n_step * 3;
        float* pl = deltas + (size_t)n * n_step * 2;
        for (uint32_t s = 0; s < n_step; ++s) {
            float dt;
            if (!march_next(&k, o, d, rd, far, &t, px, &dt)) break;
            pd[0] = d[0]; pd[1] = d[1]; pd[2] = d[2];
            pl[0] = dt; pl[1] = t - last_t; last_t = t;
            px += 3; pd += 3; pl += 2;
        }
    }
    return 0;
}

/* raymarching.cu:340-508; the global counter makes output order thread-order dependent in the
 * reference -- here rays are processed in index order (the serial emulation of _ref does the same). */
API int oracle_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                                uint32_t max_steps, uint32_t early_stop_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays,
                                int32_t* counter, const float* noises) {
    const march_k k = march_consts(bound, dt_gamma, max_steps, C, H, grid);
    for (uint32_t n = 0; n < N; ++n) {
        const float* o = rays_o + 3 * (size_t)n; const float* d = rays_d + 3 * (size_t)n;
        const float rd[3] = {1 / d[0], 1 / d[1], 1 / d[2]};
        const float near = nears[n], far = fars[n];
        float t0 = near;
        t0 += clampf(t0 * k.dt_gamma, k.dt_min, k.dt_max) * noises[n];
        float t = t0, xyz[3], dt;
        uint32_t num_steps = 0;
        while (num_steps < early_stop_steps && march_next(&k, o, d, rd, far, &t, xyz, &dt)) ++num_steps;
        const uint32_t point_index = (uint32_t)counter[0]; counter[0] += (int32_t)num_steps;
        const uint32_t ray_index = (uint32_t)counter[1]; counter[1] += 1;
        rays[3 * ray_index] = (int32_t)n; rays[3 * ray_index + 1] = (int32_t)point_index; rays[3 * ray_index + 2] = (int32_t)num_steps;
        if (num_steps == 0 || point_index + num_steps > M) continue;
        float* px = xyzs + (size_t)point_index * 3; float* pd = dirs + (size_t)point_index * 3;
        float* pl = deltas + (size_t)point_index * 2;
        t = t0;
        float last_t = near;
        for (uint32_t s = 0; s < num_steps; ++s) {
            if (!march_next(&k, o, d, rd, far, &t, px, &dt)) break;
            pd[0] = d[0]; pd[1] = d[1]; pd[2] = d[2];
            pl[0] = dt; pl[1] = t - last_t; last_t = t;
            px += 3; pd += 3; pl += 2;
        }
    }
    return 0;
}

static inline float alpha_of(float sigma, float delta, uint32_t input_alpha) {
    return input_alpha ? 0.0f + sigma : 1.0f - expf(-sigma * delta);   /* __expf on the device */
}

/* raymarching.cu:957-1046 */
API int oracle_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, uint32_t accum_deltas, uint32_t input_alpha,
                              int32_t* rays_alive, float* rays_t, const float* sigmas, const float* rgbs, const float* deltas,
                              float* weights_sum, float* depth, float* image) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)n_alive; ++n) {
        const uint32_t id = (uint32_t)rays_alive[n];
        const float* ps = sigmas + (size_t)n * n_step; const float* pc = rgbs + (size_t)n * n_step * 3;
        const float* pl = deltas + (size_t)n * n_step * 2;
        float t = rays_t[id], ws = weights_sum[id], d = depth[id];
        float r = image[3 * (size_t)id], g = image[3 * (size_t)id + 1], b = image[3 * (size_t)id + 2];
        uint32_t s = 0;
        while (s < n_step) {
            if (pl[0] == 0) break;
            const float alpha = alpha_of(ps[0], pl[0], input_alpha);
            const float T = 1 - ws;
            const float w = alpha * T;
            ws += w;
            t = accum_deltas ? t + pl[1] : pl[1];
            d += w * t;
            r += w * pc[0]; g += w * pc[1]; b += w * pc[2];
            if (T < T_thresh) break;
            ps++; pc += 3; pl += 2; s++;
        }
        if (s < n_step) rays_alive[n] = -1; else rays_t[id] = t;
        weights_sum[id] = ws; depth[id] = d;
        image[3 * (size_t)id] = r; image[3 * (size_t)id + 1] = g; image[3 * (size_t)id + 2] = b;
    }
    return 0;
}

/* raymarching.cu:529-701 */
API int oracle_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                            uint32_t M, uint32_t N, float T_thresh, uint32_t accum_deltas, uint32_t input_alpha,
                                            float* weights_sum, float* depth, float* image, float* weights) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; ++n) {
        const uint32_t id = rays[3 * n], off = rays[3 * n + 1], cnt = rays[3 * n + 2];
        float r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0, T = 1.0f;
        if (cnt != 0 && off + cnt <= M) {
            for (uint32_t s = 0; s < cnt; ++s) {
                const size_t m = (size_t)off + s;
                const float alpha = alpha_of(sigmas[m], deltas[2 * m], input_alpha);
                const float w = alpha * T;
                if (weights) weights[m] = w;
                r += w * rgbs[3 * m]; g += w * rgbs[3 * m + 1]; b += w * rgbs[3 * m + 2];
                t = accum_deltas ? t + deltas[2 * m + 1] : deltas[2 * m + 1];
                d += w * t;
                ws += w;
                T *= 1.0f - alpha;
                if (T < T_thresh) break;
            }
        }
        weights_sum[id] = ws; depth[id] = d;
        image[3 * (size_t)id] = r; image[3 * (size_t)id + 1] = g; image[3 * (size_t)id + 2] = b;
    }
    return 0;
}

/* raymarching.cu:731-821 (depth / grad_depth read at element 0: reference behaviour, kept) */
API int oracle_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* grad_depth,
                                             const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                             const float* weights_sum, const float* image, const float* depth, uint32_t M,
                                             uint32_t N, float T_thresh, float* grad_sigmas, float* grad_rgbs,
                                             uint32_t accum_deltas, uint32_t input_alpha) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; ++n) {
        const uint32_t id = rays[3 * n], off = rays[3 * n + 1], cnt = rays[3 * n + 2];
        if (cnt == 0 || off + cnt > M) continue;
        const float* gi = grad_image + 3 * (size_t)id;
        const float r_final = image[3 * (size_t)id], g_final = image[3 * (size_t)id + 1], b_final = image[3 * (size_t)id + 2];
        const float ws_final = weights_sum[id], d_final = depth[0];
        float T = 1.0f, r = 0, g = 0, b = 0, t = 0, d = 0;
        for (uint32_t s = 0; s < cnt; ++s) {
            const size_t m = (size_t)off + s;
            const float alpha = alpha_of(sigmas[m], deltas[2 * m], input_alpha);
            const float w = alpha * T;
            const float grad_scale = input_alpha ? (1.0f / (1.0f - alpha + 1e-4f)) : deltas[2 * m];
            r += w * rgbs[3 * m]; g += w * rgbs[3 * m + 1]; b += w * rgbs[3 * m + 2];
            t = accum_deltas ? t + deltas[2 * m + 1] : deltas[2 * m + 1];
            d += w * t;
            T *= 1.0f - alpha;
            grad_rgbs[3 * m] = gi[0] * w; grad_rgbs[3 * m + 1] = gi[1] * w; grad_rgbs[3 * m + 2] = gi[2] * w;
            grad_sigmas[m] = grad_scale * (gi[0] * (T * rgbs[3 * m] - (r_final - r)) + gi[1] * (T * rgbs[3 * m + 1] - (g_final - g)) +
                                           gi[2] * (T * rgbs[3 * m + 2] - (b_final - b)) + grad_depth[0] * (T * t - (d_final - d)) +
                                           grad_weights_sum[id] * (1 - ws_final));
            if (T < T_thresh) break;
        }
    }
    return 0;
}

/* the loop's `rays_alive = rays_alive[rays_alive >= 0]` (nerf/render_func/cuda_ray.py:345) */
API int oracle_compact_alive(uint32_t n_alive, const int32_t* rays_alive, int32_t* out_alive, int32_t* out_count) {
    int32_t c = 0;
    for (uint32_t n = 0; n < n_alive; ++n) if (rays_alive[n] >= 0) out_alive[c++] = rays_alive[n];
    *out_count = c;
    return 0;
}

/* ============================ feature grids ============================ */

#define MAXD 5
#define MAXC 8

typedef struct { uint32_t size, res_step; int hashed_allowed; } level_t;

/* get_grid_index (hashencoder.cu:55-70 with stride `resolution`; gridencoder.cu:54-72 with
 * resolution+1 / resolution and gridtype) */
static uint32_t grid_row(const uint32_t* p, uint32_t D, uint32_t size, uint32_t stride_step, int allow_hash) {
    static const uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= size; ++d) { index += p[d] * stride; stride *= stride_step; }
    if (allow_hash && stride > size) {
        index = 0;
        for (uint32_t d = 0; d < D; ++d) index ^= p[d] * primes[d];
    }
    return index % size;
}

/* one (point, level): kernel_grid body.  smooth=1: hashencoder.cu:103-254; smooth=0: gridencoder.cu:75-223 */
static void grid_point_level(const float* x, const float* table, uint32_t D, uint32_t C, uint32_t size, float scale,
                             uint32_t resolution, int smooth, float offset, uint32_t stride_step, int allow_hash,
                             float* out, float* dydx /* [D][C] or NULL */) {
    float pos[MAXD], dpos[MAXD];
    uint32_t cell[MAXD];
    for (uint32_t d = 0; d < D; ++d) {
        pos[d] = x[d] * scale + offset;
        cell[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)cell[d];
        if (smooth) {
            dpos[d] = 6 * pos[d] * (1.0f - pos[d]);
            pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
        } else dpos[d] = 1.0f;
    }
    (void)resolution;
    for (uint32_t c = 0; c < C; ++c) out[c] = 0;
    for (uint32_t idx = 0; idx < (1u << D); ++idx) {
        float w = 1;
        uint32_t q[MAXD];
        for (uint32_t d = 0; d < D; ++d) {
            if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; q[d] = cell[d]; }
            else { w *= pos[d]; q[d] = cell[d] + 1; }
        }
        const uint32_t row = grid_row(q, D, size, stride_step, allow_hash);
        for (uint32_t c = 0; c < C; ++c) out[c] += w * table[(size_t)row * C + c];
    }
    if (!dydx) return;
    for (uint32_t gd = 0; gd < D; ++gd) {
        float acc[MAXC];
        for (uint32_t c = 0; c < C; ++c) acc[c] = 0;
        for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
            float w = scale;
            uint32_t q[MAXD];
            for (uint32_t nd = 0; nd < D - 1; ++nd) {
                const uint32_t d = nd >= gd ? nd + 1 : nd;
                if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; q[d] = cell[d]; }
                else { w *= pos[d]; q[d] = cell[d] + 1; }
            }
            q[gd] = cell[gd];
            const uint32_t left = grid_row(q, D, size, stride_step, allow_hash);
            q[gd] = cell[gd] + 1;
            const uint32_t right = grid_row(q, D, size, stride_step, allow_hash);
            for (uint32_t c = 0; c < C; ++c) {
                if (smooth) acc[c] += w * (table[(size_t)right * C + c] - table[(size_t)left * C + c]) * dpos[gd];
                else acc[c] += w * (table[(size_t)right * C + c] - table[(size_t)left * C + c]);
            }
        }
        for (uint32_t c = 0; c < C; ++c) dydx[gd * C + c] = acc[c];
    }
}

static int grid_forward(const float* inputs, const float* emb, const int32_t* offsets, float* outputs, uint32_t B, uint32_t D,
                        uint32_t C, uint32_t L, float S, uint32_t H, float* dy_dx, int smooth, uint32_t gridtype,
                        int align_corners) {
    if (D < 1 || D > MAXD || C < 1 || C > MAXC) return -1;
    for (uint32_t l = 0; l < L; ++l) {
        const float scale = exp2f(l * S) * H - 1.0f;
        const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
        const uint32_t size = (uint32_t)(offsets[l + 1] - offsets[l]);
        const float* table = emb + (size_t)offsets[l] * C;
        const float offset = smooth ? 0.0f : (align_corners ? 0.0f : 0.5f);
        const uint32_t step = smooth ? resolution : (align_corners ? resolution : resolution + 1);
        const int allow_hash = smooth ? 1 : (gridtype == 0);
#pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < (int64_t)B; ++b) {
            const float* x = inputs + (size_t)b * D;
            float* out = outputs + ((size_t)l * B + b) * C;
            float* g = dy_dx ? dy_dx + ((size_t)b * L + l) * D * C : NULL;
            int oob = 0;
            for (uint32_t d = 0; d < D; ++d) if (x[d] < 0 || x[d] > 1) oob = 1;
            if (oob) {
                for (uint32_t c = 0; c < C; ++c) out[c] = 0;
                if (g) for (uint32_t i = 0; i < D * C; ++i) g[i] = 0;
                continue;
            }
            grid_point_level(x, table, D, C, size, scale, resolution, smooth, offset, step, allow_hash, out, g);
        }
    }
    return 0;
}

/* hashencoder.cu:725-760 */
API int oracle_hash_encode_forward(const float* inputs, const float* embeddings, const int32_t* offsets, float* outputs,
                                   uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs,
                                   float* dy_dx) {
    if (D != 2 && D != 3) return -1;
    return grid_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc_grad_inputs ? dy_dx : NULL, 1, 0, 0);
}
/* The table rows kernel_grid<float,3,C> reads for one point: for every level the eight corners
 * pos_grid_local = cell + (bit d of idx) (hashencoder.cu:175-190), each through get_grid_index (:55-70) =
 * grid_row() above; rows_out [B, L, 8], corner index = idx (bit d <-> dimension d).  Points outside [0,1]^3 read nothing
 * (:124-149): their rows are written as 0xffffffff.  Test hook for the index arithmetic of GPU gather paths. */
API int oracle_hash_corner_rows(const float* inputs, const int32_t* offsets, uint32_t* rows_out, uint32_t B, uint32_t L, float S,
                                uint32_t H) {
    const uint32_t D = 3;
    for (uint32_t l = 0; l < L; ++l) {
        const float scale = exp2f(l * S) * H - 1.0f;
        const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
        const uint32_t size = (uint32_t)(offsets[l + 1] - offsets[l]);
#pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < (int64_t)B; ++b) {
            const float* x = inputs + (size_t)b * D;
            uint32_t* out = rows_out + ((size_t)b * L + l) * 8;
            int oob = 0;
            for (uint32_t d = 0; d < D; ++d) if (x[d] < 0 || x[d] > 1) oob = 1;
            if (oob) { for (uint32_t i = 0; i < 8; ++i) out[i] = 0xffffffffu; continue; }
            uint32_t cell[3];
            for (uint32_t d = 0; d < D; ++d) cell[d] = (uint32_t)floorf(x[d] * scale);
            for (uint32_t idx = 0; idx < 8; ++idx) {
                uint32_t q[3];
                for (uint32_t d = 0; d < D; ++d) q[d] = cell[d] + ((idx >> d) & 1u);
                out[idx] = grid_row(q, D, size, resolution, 1);
            }
        }
    }
    return 0;
}

/* gridencoder.cu:423-450 */
API int oracle_grid_encode_forward(const float* inputs, const float* embeddings, const int32_t* offsets, float* outputs,
                                   uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, float* dy_dx,
                                   uint32_t gridtype, int align_corners) {
    return grid_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, 0, gridtype, align_corners);
}

/* table scatter + input gradient; serial over points so the fp32 accumulation order is the
 * reference's thread order under the serial emulation (hashencoder.cu:257-372, gridencoder.cu:226-340) */
static int grid_backward(const float* grad, const float* inputs, const int32_t* offsets, float* grad_emb, uint32_t B,
                         uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const float* dy_dx, float* grad_inputs,
                         int smooth, uint32_t gridtype, int align_corners) {
    if (D < 1 || D > MAXD || C < 1 || C > MAXC) return -1;
    if (grad_emb) {
        for (uint32_t l = 0; l < L; ++l) {
            const float scale = exp2f(l * S) * H - 1.0f;
            const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
            const uint32_t size = (uint32_t)(offsets[l + 1] - offsets[l]);
            float* table = grad_emb + (size_t)offsets[l] * C;
            const float offset = smooth ? 0.0f : (align_corners ? 0.0f : 0.5f);
            const uint32_t step = smooth ? resolution : (align_corners ? resolution : resolution + 1);
            const int allow_hash = smooth ? 1 : (gridtype == 0);
            for (uint32_t b = 0; b < B; ++b) {
                const float* x = inputs + (size_t)b * D;
                int oob = 0;
                for (uint32_t d = 0; d < D; ++d) if (x[d] < 0 || x[d] > 1) oob = 1;
                if (oob) continue;
                float pos[MAXD]; uint32_t cell[MAXD];
                for (uint32_t d = 0; d < D; ++d) {
                    pos[d] = x[d] * scale + offset;
                    cell[d] = (uint32_t)floorf(pos[d]);
                    pos[d] -= (float)cell[d];
                    if (smooth) pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
                }
                /* the reference handles N_C = min(2,C) channels per thread, threads ordered by channel group */
                const uint32_t NC = C < 2 ? C : 2;
                for (uint32_t ch = 0; ch < C; ch += NC) {
                    for (uint32_t idx = 0; idx < (1u << D); ++idx) {
                        float w = 1; uint32_t q[MAXD];
                        for (uint32_t d = 0; d < D; ++d) {
                            if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; q[d] = cell[d]; }
                            else { w *= pos[d]; q[d] = cell[d] + 1; }
                        }
                        const uint32_t row = grid_row(q, D, size, step, allow_hash);
                        for (uint32_t c = 0; c < NC; ++c)
                            table[(size_t)row * C + ch + c] += w * grad[((size_t)l * B + b) * C + ch + c];
                    }
                }
            }
        }
    }
    if (dy_dx && grad_inputs) {
#pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < (int64_t)B * D; ++t) {
            const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (int64_t)b * D);
            float acc = 0;
            for (uint32_t l = 0; l < L; ++l)
                for (uint32_t c = 0; c < C; ++c)
                    acc += grad[((size_t)l * B + b) * C + c] * dy_dx[(size_t)b * L * D * C + (size_t)l * D * C + d * C + c];
            grad_inputs[t] = acc;
        }
    }
    return 0;
}
API int oracle_hash_encode_backward(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets,
                                    float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                    int calc_grad_inputs, const float* dy_dx, float* grad_inputs) {
    (void)embeddings;
    if (D != 2 && D != 3) return -1;
    return grid_backward(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, calc_grad_inputs ? dy_dx : NULL,
                         grad_inputs, 1, 0, 0);
}
API int oracle_grid_encode_backward(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets,
                                    float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                    const float* dy_dx, float* grad_inputs, uint32_t gridtype, int align_corners) {
    (void)embeddings;
    return grid_backward(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, 0, gridtype, align_corners);
}

/* hashencoder.cu:375-595 */
API int oracle_hash_encode_second_backward(const float* grad, const float* inputs, const float* embeddings,
                                           const int32_t* offsets, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                           uint32_t H, int calc_grad_inputs, const float* dy_dx, const float* ggx,
                                           float* grad_grad, float* grad2_emb) {
    (void)embeddings; (void)calc_grad_inputs;
    if ((D != 2 && D != 3) || C == 1 || C > MAXC) return -1;
    for (uint32_t l = 0; l < L; ++l)
        for (uint32_t b = 0; b < B; ++b)
            for (uint32_t c = 0; c < C; ++c) {
                float r = 0;
                for (uint32_t d = 0; d < D; ++d) r += ggx[(size_t)b * D + d] * dy_dx[((size_t)b * L + l) * D * C + d * C + c];
                grad_grad[((size_t)l * B + b) * C + c] = r;
            }
    for (uint32_t l = 0; l < L; ++l) {
        const float scale = exp2f(l * S) * H - 1.0f;
        const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
        const uint32_t size = (uint32_t)(offsets[l + 1] - offsets[l]);
        float* table = grad2_emb + (size_t)offsets[l] * C;
        for (uint32_t b = 0; b < B; ++b) {
            const float* x = inputs + (size_t)b * D;
            int oob = 0;
            for (uint32_t d = 0; d < D; ++d) if (x[d] < 0 || x[d] > 1) oob = 1;
            if (oob) continue;
            float pos[MAXD], dpos[MAXD]; uint32_t cell[MAXD];
            for (uint32_t d = 0; d < D; ++d) {
                pos[d] = x[d] * scale;
                cell[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)cell[d];
                dpos[d] = 6 * pos[d] * (1.0f - pos[d]);
                pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
            }
            for (uint32_t ch = 0; ch < C; ch += 2) {   /* N_C = 2 channels per reference thread */
                float cache[8][2];
                memset(cache, 0, sizeof(cache));
                for (uint32_t gd = 0; gd < D; ++gd)
                    for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
                        float w = scale; uint32_t lo = 0;
                        for (uint32_t nd = 0; nd < D - 1; ++nd) {
                            const uint32_t d = nd >= gd ? nd + 1 : nd;
                            if ((idx & (1u << nd)) == 0) w *= 1 - pos[d];
                            else { w *= pos[d]; lo |= 1u << d; }
                        }
                        const uint32_t hi = lo | (1u << gd);
                        for (uint32_t c = 0; c < 2; ++c) {
                            const float v = w * grad[((size_t)l * B + b) * C + ch + c] * ggx[(size_t)b * D + gd] * dpos[gd];
                            cache[hi][c] += v;
                            cache[lo][c] -= v;
                        }
                    }
                for (uint32_t idx = 0; idx < (1u << D); ++idx) {
                    uint32_t q[MAXD];
                    for (uint32_t d = 0; d < D; ++d) q[d] = cell[d] + ((idx >> d) & 1u);
                    const uint32_t row = grid_row(q, D, size, resolution, 1);
                    for (uint32_t c = 0; c < 2; ++c) table[(size_t)row * C + ch + c] += cache[idx][c];
                }
            }
        }
    }
    return 0;
}

/* ============================ frequency encoding ============================ */
/* freqencoder.cu:30-58: cos is evaluated as sin(x + pi/2) in fp32 */
API int oracle_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs) {
    (void)deg;
    const float half_pi = 3.141592653589793f / 2;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < (int64_t)B * C; ++t) {
        const uint32_t b = (uint32_t)(t / C), c = (uint32_t)(t - (int64_t)b * C);
        const float* x = inputs + (size_t)b * D;
        if (c < D) outputs[t] = x[c];
        else {
            const uint32_t col = c / D - 1, d = c % D, freq = col / 2;
            const float phase = (col % 2) * half_pi;
            outputs[t] = sinf(scalbnf(x[d], (int)freq) + phase);
        }
    }
    return 0;
}
/* freqencoder.cu:63-94 */
API int oracle_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                                    float* grad_inputs) {
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < (int64_t)B * D; ++t) {
        const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (int64_t)b * D);
        const float* g = grad + (size_t)b * C; const float* o = outputs + (size_t)b * C;
        float r = g[d];
        g += D; o += D;
        for (uint32_t f = 0; f < deg; ++f) {
            r += scalbnf(1.0f, (int)f) * (g[d] * o[D + d] - g[D + d] * o[d]);
            g += 2 * D; o += 2 * D;
        }
        grad_inputs[t] = r;
    }
    return 0;
}

/* ============================ spherical harmonics ============================ */
/* The reference (shencoder.cu:46-121, 130-356) hard-codes each basis function as a polynomial
 * c * Q_l^m(z) * {Re,Im}((x+iy)^|m|) and its symbolic x/y/z derivatives.  The oracle evaluates the
 * same polynomials from their definition, in double:
 *   Y_l^m = K_l^|m| * Q_l^|m|(z) * sqrt(2)^(m!=0) * (m>=0 ? Re : Im)((x+iy)^|m|),
 *   Q_l^m(z) = d^m/dz^m P_l(z)  with the Condon-Shortley sign (-1)^m,
 *   K_l^m = sqrt((2l+1)/(4 pi) (l-m)!/(l+m)!)
 * and rounds once to fp32.  Ordering: index l*l + l + m, m = -l..l. */
static void legendre_coeffs(int l, double* c /* [l+1], P_l(z) = sum c[k] z^k */) {
    double p0[16] = {1}, p1[16] = {0, 1}, p2[16];
    if (l == 0) { c[0] = 1; return; }
    for (int n = 1; n < l; ++n) {           /* (n+1) P_{n+1} = (2n+1) z P_n - n P_{n-1} */
        memset(p2, 0, sizeof(p2));
        for (int k = 0; k <= n; ++k) p2[k + 1] += (2.0 * n + 1) * p1[k] / (n + 1);
        for (int k = 0; k <= n - 1; ++k) p2[k] -= (double)n * p0[k] / (n + 1);
        memcpy(p0, p1, sizeof(p0)); memcpy(p1, p2, sizeof(p1));
    }
    for (int k = 0; k <= l; ++k) c[k] = p1[k];
}
static double factorial(int n) { double f = 1; for (int i = 2; i <= n; ++i) f *= i; return f; }

API int oracle_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C, float* dy_dx) {
    if (D != 3 || C < 1 || C > 8) return -1;
    const uint32_t C2 = C * C;
    /* Q[l][m][k]: coefficient of z^k in (-1)^m d^m/dz^m P_l, times the normalisation */
    static double Q[8][8][9];
    for (int l = 0; l < (int)C; ++l) {
        double pl[16];
        legendre_coeffs(l, pl);
        for (int m = 0; m <= l; ++m) {
            double d[16];
            memcpy(d, pl, sizeof(double) * (l + 1));
            int deg = l;
            for (int j = 0; j < m; ++j) { for (int k = 0; k < deg; ++k) d[k] = d[k + 1] * (k + 1); deg--; }
            const double K = sqrt((2.0 * l + 1) / (4 * M_PI) * factorial(l - m) / factorial(l + m));
            const double s = (m % 2 ? -1.0 : 1.0) * K * (m ? sqrt(2.0) : 1.0);
            for (int k = 0; k <= 8; ++k) Q[l][m][k] = k <= deg ? s * d[k] : 0.0;
        }
    }
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)B; ++b) {
        const double x = inputs[3 * b], y = inputs[3 * b + 1], z = inputs[3 * b + 2];
        double re[9], im[9];   /* (x+iy)^m */
        re[0] = 1; im[0] = 0;
        for (int m = 1; m <= 8; ++m) { re[m] = re[m - 1] * x - im[m - 1] * y; im[m] = re[m - 1] * y + im[m - 1] * x; }
        for (int l = 0; l < (int)C; ++l)
            for (int m = -l; m <= l; ++m) {
                const int am = m < 0 ? -m : m;
                double q = 0, dq = 0;   /* Q(z), Q'(z) */
                for (int k = 8; k >= 0; --k) q = q * z + Q[l][am][k];
                for (int k = 8; k >= 1; --k) dq = dq * z + Q[l][am][k] * k;
                const double ang = m >= 0 ? re[am] : im[am];
                /* d/dx (x+iy)^m = m (x+iy)^(m-1);  d/dy (x+iy)^m = i m (x+iy)^(m-1) */
                const double dre_dx = am ? am * re[am - 1] : 0, dim_dx = am ? am * im[am - 1] : 0;
                const double dre_dy = am ? -am * im[am - 1] : 0, dim_dy = am ? am * re[am - 1] : 0;
                const uint32_t idx = (uint32_t)(l * l + l + m);
                outputs[(size_t)b * C2 + idx] = (float)(q * ang);
                if (dy_dx) {
                    float* g = dy_dx + (size_t)b * 3 * C2;
                    g[idx] = (float)(q * (m >= 0 ? dre_dx : dim_dx));
                    g[C2 + idx] = (float)(q * (m >= 0 ? dre_dy : dim_dy));
                    g[2 * C2 + idx] = (float)(dq * ang);
                }
            }
    }
    return 0;
}
/* shencoder.cu:359-379: accumulates into grad_inputs */
API int oracle_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C, const float* dy_dx,
                                  float* grad_inputs) {
    (void)inputs;
    const uint32_t C2 = C * C;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < (int64_t)B * D; ++t) {
        const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (int64_t)b * D);
        for (uint32_t ch = 0; ch < C2; ++ch) grad_inputs[t] += grad[(size_t)b * C2 + ch] * dy_dx[(size_t)b * D * C2 + d * C2 + ch];
    }
    return 0;
}

/* ============================ feature grids, half precision ============================ */
/* The reference dispatches its grid kernels on at::Half too (AT_DISPATCH_FLOATING_TYPES_AND_HALF: hashencoder.cu:747,778;
 * gridencoder.cu:443,474): `scalar_t` = at::Half for the table, the outputs, dy_dx and the gradients (hashencoder: also the
 * inputs; gridencoder keeps `const float* inputs`).  c10::Half does ALL arithmetic in float and narrows (round to nearest
 * even) only where a value becomes a Half again (c10/util/Half-inl.h):
 *     Half (+,-,*) Half -> Half(float(a) op float(b))            e.g. `grid[r] - grid[l]`, `grad * dy_dx`
 *     float op Half     -> float                                  e.g. `w * grid[i]`, `w * grad_cur[c]`
 *     Half += float     -> a = Half(float(a) + float(Half(b)))   (operator+=(Half&, const Half&): the float narrows FIRST)
 * The functions below restate the kernels with those narrowing points written out (H() = narrow, F() = widen). */
static uint16_t f2h(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
    if (x < 0x38800000u) {
        if (x < 0x33000000u) return (uint16_t)sign;
        const int e = (int)(x >> 23);
        const uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const int shift = 126 - e;
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = (x - 0x38000000u) >> 13;
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
    return (uint16_t)(sign | r);
}
static float h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { const float f = (float)m * 5.9604644775390625e-08f; memcpy(&x, &f, 4); x |= sign; }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &x, 4); return f;
}
API uint16_t oracle_float_to_half(float f) { return f2h(f); }
API float oracle_half_to_float(uint16_t h) { return h2f(h); }
#define H(x) f2h(x)
#define F(x) h2f(x)
static inline uint16_t h_add_f(uint16_t a, float b) { return H(F(a) + F(H(b))); }      /* Half += float */

/* one (point, level) of kernel_grid<at::Half, D, C>: x in float (hashencoder: widened from its Half inputs, `(float)inputs[d]`) */
static void grid_point_level_h(const float* x, const uint16_t* table, uint32_t D, uint32_t C, uint32_t size, float scale, int smooth,
                               float offset, uint32_t stride_step, int allow_hash, uint16_t* out, uint16_t* dydx /* [D][C] or NULL */) {
    float pos[MAXD], dpos[MAXD];
    uint32_t cell[MAXD];
    for (uint32_t d = 0; d < D; ++d) {
        pos[d] = x[d] * scale + offset;
        cell[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)cell[d];
        if (smooth) { dpos[d] = 6 * pos[d] * (1.0f - pos[d]); pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]); }
        else dpos[d] = 1.0f;
    }
    uint16_t res[MAXC];
    for (uint32_t c = 0; c < C; ++c) res[c] = H(0.0f);
    for (uint32_t idx = 0; idx < (1u << D); ++idx) {
        float w = 1;
        uint32_t q[MAXD];
        for (uint32_t d = 0; d < D; ++d) {
            if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; q[d] = cell[d]; }
            else { w *= pos[d]; q[d] = cell[d] + 1; }
        }
        const uint32_t row = grid_row(q, D, size, stride_step, allow_hash);
        for (uint32_t c = 0; c < C; ++c) res[c] = h_add_f(res[c], w * F(table[(size_t)row * C + c]));    /* results[ch] += w * grid[..] */
    }
    for (uint32_t c = 0; c < C; ++c) out[c] = res[c];
    if (!dydx) return;
    for (uint32_t gd = 0; gd < D; ++gd) {
        uint16_t acc[MAXC];
        for (uint32_t c = 0; c < C; ++c) acc[c] = H(0.0f);
        for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
            float w = scale;
            uint32_t q[MAXD];
            for (uint32_t nd = 0; nd < D - 1; ++nd) {
                const uint32_t d = nd >= gd ? nd + 1 : nd;
                if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; q[d] = cell[d]; }
                else { w *= pos[d]; q[d] = cell[d] + 1; }
            }
            q[gd] = cell[gd];
            const uint32_t left = grid_row(q, D, size, stride_step, allow_hash);
            q[gd] = cell[gd] + 1;
            const uint32_t right = grid_row(q, D, size, stride_step, allow_hash);
            for (uint32_t c = 0; c < C; ++c) {
                const uint16_t diff = H(F(table[(size_t)right * C + c]) - F(table[(size_t)left * C + c]));     /* Half - Half -> Half */
                const float t = smooth ? w * F(diff) * dpos[gd] : w * F(diff);
                acc[c] = h_add_f(acc[c], t);
            }
        }
        for (uint32_t c = 0; c < C; ++c) dydx[gd * C + c] = acc[c];
    }
}

static int grid_forward_h(const void* inputs, int inputs_half, const uint16_t* emb, const int32_t* offsets, uint16_t* outputs, uint32_t B,
                          uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H_, uint16_t* dy_dx, int smooth, uint32_t gridtype,
                          int align_corners) {
    if (D < 1 || D > MAXD || C < 1 || C > MAXC) return -1;
    for (uint32_t l = 0; l < L; ++l) {
        const float scale = exp2f(l * S) * H_ - 1.0f;
        const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
        const uint32_t size = (uint32_t)(offsets[l + 1] - offsets[l]);
        const uint16_t* table = emb + (size_t)offsets[l] * C;
        const float offset = smooth ? 0.0f : (align_corners ? 0.0f : 0.5f);
        const uint32_t step = smooth ? resolution : (align_corners ? resolution : resolution + 1);
        const int allow_hash = smooth ? 1 : (gridtype == 0);
#pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < (int64_t)B; ++b) {
            float x[MAXD];
            for (uint32_t d = 0; d < D; ++d)
                x[d] = inputs_half ? F(((const uint16_t*)inputs)[(size_t)b * D + d]) : ((const float*)inputs)[(size_t)b * D + d];
            uint16_t* out = outputs + ((size_t)l * B + b) * C;
            uint16_t* g = dy_dx ? dy_dx + ((size_t)b * L + l) * D * C : NULL;
            int oob = 0;
            for (uint32_t d = 0; d < D; ++d) if (x[d] < 0 || x[d] > 1) oob = 1;
            if (oob) {
                for (uint32_t c = 0; c < C; ++c) out[c] = H(0.0f);
                if (g) for (uint32_t i = 0; i < D * C; ++i) g[i] = H(0.0f);
                continue;
            }
            grid_point_level_h(x, table, D, C, size, scale, smooth, offset, step, allow_hash, out, g);
        }
    }
    return 0;
}

/* kernel_grid_backward<at::Half,..> (paired __half2 atomics: each w * grad narrowed, then an fp16 add per component) +
 * kernel_input_backward<at::Half,..> (Half product, Half running sum); serial over points like grid_backward() */
static int grid_backward_h(const uint16_t* grad, const void* inputs, int inputs_half, const int32_t* offsets, uint16_t* grad_emb, uint32_t B,
                           uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H_, const uint16_t* dy_dx, uint16_t* grad_inputs, int smooth,
                           uint32_t gridtype, int align_corners) {
    if (D < 1 || D > MAXD || C < 1 || C > MAXC) return -1;
    if (grad_emb) {
        for (uint32_t l = 0; l < L; ++l) {
            const float scale = exp2f(l * S) * H_ - 1.0f;
            const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
            const uint32_t size = (uint32_t)(offsets[l + 1] - offsets[l]);
            uint16_t* table = grad_emb + (size_t)offsets[l] * C;
            const float offset = smooth ? 0.0f : (align_corners ? 0.0f : 0.5f);
            const uint32_t step = smooth ? resolution : (align_corners ? resolution : resolution + 1);
            const int allow_hash = smooth ? 1 : (gridtype == 0);
            for (uint32_t b = 0; b < B; ++b) {
                float x[MAXD];
                int oob = 0;
                for (uint32_t d = 0; d < D; ++d) {
                    x[d] = inputs_half ? F(((const uint16_t*)inputs)[(size_t)b * D + d]) : ((const float*)inputs)[(size_t)b * D + d];
                    if (x[d] < 0 || x[d] > 1) oob = 1;
                }
                if (oob) continue;
                float pos[MAXD]; uint32_t cell[MAXD];
                for (uint32_t d = 0; d < D; ++d) {
                    pos[d] = x[d] * scale + offset;
                    cell[d] = (uint32_t)floorf(pos[d]);
                    pos[d] -= (float)cell[d];
                    if (smooth) pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
                }
                const uint32_t NC = C < 2 ? C : 2;
                for (uint32_t ch = 0; ch < C; ch += NC) {
                    for (uint32_t idx = 0; idx < (1u << D); ++idx) {
                        float w = 1; uint32_t q[MAXD];
                        for (uint32_t d = 0; d < D; ++d) {
                            if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; q[d] = cell[d]; }
                            else { w *= pos[d]; q[d] = cell[d] + 1; }
                        }
                        const uint32_t row = grid_row(q, D, size, step, allow_hash);
                        for (uint32_t c = 0; c < NC; ++c) {
                            uint16_t* t = &table[(size_t)row * C + ch + c];
                            const uint16_t v = H(w * F(grad[((size_t)l * B + b) * C + ch + c]));      /* (__half)(w * grad_cur[c]) */
                            *t = H(F(*t) + F(v));                                                     /* fp16 atomic add      */
                        }
                    }
                }
            }
        }
    }
    if (dy_dx && grad_inputs) {
#pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < (int64_t)B * D; ++t) {
            const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (int64_t)b * D);
            uint16_t acc = H(0.0f);
            for (uint32_t l = 0; l < L; ++l)
                for (uint32_t c = 0; c < C; ++c) {
                    const uint16_t prod = H(F(grad[((size_t)l * B + b) * C + c]) * F(dy_dx[(size_t)b * L * D * C + (size_t)l * D * C + d * C + c]));
                    acc = H(F(acc) + F(prod));
                }
            grad_inputs[t] = acc;
        }
    }
    return 0;
}
API int oracle_hash_encode_forward_f16(const uint16_t* inputs, const uint16_t* embeddings, const int32_t* offsets, uint16_t* outputs, uint32_t B,
                                       uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H_, int calc_grad_inputs, uint16_t* dy_dx) {
    if (D != 2 && D != 3) return -1;
    return grid_forward_h(inputs, 1, embeddings, offsets, outputs, B, D, C, L, S, H_, calc_grad_inputs ? dy_dx : NULL, 1, 0, 0);
}
API int oracle_hash_encode_backward_f16(const uint16_t* grad, const uint16_t* inputs, const uint16_t* embeddings, const int32_t* offsets,
                                        uint16_t* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H_,
                                        int calc_grad_inputs, const uint16_t* dy_dx, uint16_t* grad_inputs) {
    (void)embeddings;
    if (D != 2 && D != 3) return -1;
    return grid_backward_h(grad, inputs, 1, offsets, grad_embeddings, B, D, C, L, S, H_, calc_grad_inputs ? dy_dx : NULL, grad_inputs, 1, 0, 0);
}
/* hash_encode_second_backward<at::Half> (hashencoder.cu:375-428 grad_grad, :431-595 table): Half products and sums in the first
 * kernel; in the second the corner cache is Half -- `cache +-= w * grad * ggx[gd] * smoothstep'` narrows the float product, then adds /
 * subtracts in Half -- and `(__half)(1.0 * cache)` is exact; the scatter is an fp16 (pair) atomic add, serial here */
API int oracle_hash_encode_second_backward_f16(const uint16_t* grad, const uint16_t* inputs, const uint16_t* embeddings, const int32_t* offsets,
                                               uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H_, int calc_grad_inputs,
                                               const uint16_t* dy_dx, const uint16_t* ggx, uint16_t* grad_grad, uint16_t* grad2_emb) {
    (void)embeddings; (void)calc_grad_inputs;
    if ((D != 2 && D != 3) || C == 1 || C > MAXC) return -1;
    for (uint32_t l = 0; l < L; ++l)
        for (uint32_t b = 0; b < B; ++b)
            for (uint32_t c = 0; c < C; ++c) {
                uint16_t r = H(0.0f);
                for (uint32_t d = 0; d < D; ++d) r = H(F(r) + F(H(F(ggx[(size_t)b * D + d]) * F(dy_dx[((size_t)b * L + l) * D * C + d * C + c]))));
                grad_grad[((size_t)l * B + b) * C + c] = r;
            }
    for (uint32_t l = 0; l < L; ++l) {
        const float scale = exp2f(l * S) * H_ - 1.0f;
        const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
        const uint32_t size = (uint32_t)(offsets[l + 1] - offsets[l]);
        uint16_t* table = grad2_emb + (size_t)offsets[l] * C;
        for (uint32_t b = 0; b < B; ++b) {
            float x[MAXD];
            int oob = 0;
            for (uint32_t d = 0; d < D; ++d) { x[d] = F(inputs[(size_t)b * D + d]); if (x[d] < 0 || x[d] > 1) oob = 1; }
            if (oob) continue;
            float pos[MAXD], dpos[MAXD]; uint32_t cell[MAXD];
            for (uint32_t d = 0; d < D; ++d) {
                pos[d] = x[d] * scale;
                cell[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)cell[d];
                dpos[d] = 6 * pos[d] * (1.0f - pos[d]);
                pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
            }
            for (uint32_t ch = 0; ch < C; ch += 2) {
                uint16_t cache[8][2];
                for (int i = 0; i < 8; ++i) cache[i][0] = cache[i][1] = H(0.0f);
                for (uint32_t gd = 0; gd < D; ++gd)
                    for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
                        float w = scale; uint32_t lo = 0;
                        for (uint32_t nd = 0; nd < D - 1; ++nd) {
                            const uint32_t d = nd >= gd ? nd + 1 : nd;
                            if ((idx & (1u << nd)) == 0) w *= 1 - pos[d];
                            else { w *= pos[d]; lo |= 1u << d; }
                        }
                        const uint32_t hi = lo | (1u << gd);
                        for (uint32_t c = 0; c < 2; ++c) {
                            const uint16_t v = H(w * F(grad[((size_t)l * B + b) * C + ch + c]) * F(ggx[(size_t)b * D + gd]) * dpos[gd]);
                            cache[hi][c] = H(F(cache[hi][c]) + F(v));
                            cache[lo][c] = H(F(cache[lo][c]) - F(v));
                        }
                    }
                for (uint32_t idx = 0; idx < (1u << D); ++idx) {
                    uint32_t q[MAXD];
                    for (uint32_t d = 0; d < D; ++d) q[d] = cell[d] + ((idx >> d) & 1u);
                    const uint32_t row = grid_row(q, D, size, resolution, 1);
                    for (uint32_t c = 0; c < 2; ++c) { uint16_t* t = &table[(size_t)row * C + ch + c]; *t = H(F(*t) + F(cache[idx][c])); }
                }
            }
        }
    }
    return 0;
}
/* shencoder.cu:413,435 on at::Half.  Forward: the exact basis (the double evaluation above) of the widened direction, rounded to
 * fp32 and then to fp16 -- NOT the reference's half arithmetic, which rounds every monomial (x2 = H(x*x), x4 = H(x2*x2) ...) and so
 * sits a few fp16 ulp from the exact basis; tests/test_oracle_pinning.py measures that distance against the reference's own
 * template.  Backward: `grad_inputs[t] += grad[ch] * dy_dx[ch]` in Half, term by term, as the reference (shencoder.cu:359-379). */
API int oracle_sh_encode_forward_f16(const uint16_t* inputs, uint16_t* outputs, uint32_t B, uint32_t D, uint32_t C, uint16_t* dy_dx) {
    if (D != 3 || C < 1 || C > 8) return -1;
    const uint32_t C2 = C * C;
    float* in = (float*)malloc(sizeof(float) * 3 * (size_t)(B ? B : 1));
    float* out = (float*)malloc(sizeof(float) * C2 * (size_t)(B ? B : 1));
    float* dy = dy_dx ? (float*)malloc(sizeof(float) * 3 * C2 * (size_t)(B ? B : 1)) : NULL;
    for (size_t i = 0; i < (size_t)B * 3; ++i) in[i] = F(inputs[i]);
    const int rc = oracle_sh_encode_forward(in, out, B, D, C, dy);
    for (size_t i = 0; i < (size_t)B * C2; ++i) outputs[i] = H(out[i]);
    if (dy) for (size_t i = 0; i < (size_t)B * 3 * C2; ++i) dy_dx[i] = H(dy[i]);
    free(in); free(out); free(dy);
    return rc;
}
API int oracle_sh_encode_backward_f16(const uint16_t* grad, const uint16_t* inputs, uint32_t B, uint32_t D, uint32_t C, const uint16_t* dy_dx,
                                      uint16_t* grad_inputs) {
    (void)inputs;
    const uint32_t C2 = C * C;
    for (int64_t t = 0; t < (int64_t)B * D; ++t) {
        const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (int64_t)b * D);
        uint16_t acc = grad_inputs[t];
        for (uint32_t ch = 0; ch < C2; ++ch) acc = H(F(acc) + F(H(F(grad[(size_t)b * C2 + ch]) * F(dy_dx[(size_t)b * D * C2 + d * C2 + ch]))));
        grad_inputs[t] = acc;
    }
    return 0;
}
API int oracle_grid_encode_forward_f16(const float* inputs, const uint16_t* embeddings, const int32_t* offsets, uint16_t* outputs, uint32_t B,
                                       uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H_, uint16_t* dy_dx, uint32_t gridtype, int align_corners) {
    return grid_forward_h(inputs, 0, embeddings, offsets, outputs, B, D, C, L, S, H_, dy_dx, 0, gridtype, align_corners);
}
API int oracle_grid_encode_backward_f16(const uint16_t* grad, const float* inputs, const uint16_t* embeddings, const int32_t* offsets,
                                        uint16_t* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H_,
                                        const uint16_t* dy_dx, uint16_t* grad_inputs, uint32_t gridtype, int align_corners) {
    (void)embeddings;
    return grid_backward_h(grad, inputs, 0, offsets, grad_embeddings, B, D, C, L, S, H_, dy_dx, grad_inputs, 0, gridtype, align_corners);
}
#undef H
#undef F

/* ============================ integrated directional encoding ============================ */
/* ide_encoder/ide_encoder.py:5-55 (coefficient tables) and :98-130 (forward).  Evaluated in double
 * from the closed forms on the reference's fp32-rounded coefficient table, rounded once: this is the
 * reference's polynomial evaluated WITHOUT its fp32 cancellation noise (up to ~1e-2 for l = 16 near
 * |z| = 1, DESIGN.md "IDE numerics").  The reference's actual fp32 outputs are pinned by golden
 * vectors generated from the imported module (tests/golden/make_golden.py). */
static double gen_binom(double a, int k) { double p = 1; for (int i = 0; i < k; ++i) p *= (a - i); return p / factorial(k); }
static double ide_coeff(int l, int m, int k) {   /* sph_harm_coeff(l, m, k) */
    const double al = (m % 2 ? -1.0 : 1.0) * pow(2.0, l) * factorial(l) / factorial(k) / factorial(l - k - m) *
                      gen_binom(0.5 * (l + k + m - 1.0), l);
    return sqrt((2.0 * l + 1.0) * factorial(l - m) / (4.0 * M_PI * factorial(l + m))) * al;
}
API int oracle_ide_encode_forward(const float* dirs, const float* roughness_ptr, float roughness_scalar, uint32_t B,
                                  uint32_t deg_view, float* outputs) {
    if (deg_view < 1 || deg_view > 5) return -1;
    int ml_m[64], ml_l[64], n = 0;
    for (uint32_t i = 0; i < deg_view; ++i) { const int l = 1 << i; for (int m = 0; m <= l; ++m) { ml_m[n] = m; ml_l[n] = l; ++n; } }
    const int lmax = 1 << (deg_view - 1);
    double* mat = (double*)calloc((size_t)(lmax + 1) * n, sizeof(double));
    for (int i = 0; i < n; ++i)
        for (int k = 0; k <= ml_l[i] - ml_m[i]; ++k)
            mat[(size_t)k * n + i] = (double)(float)ide_coeff(ml_l[i], ml_m[i], k);   /* torch.Tensor(mat): fp32 table */
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)B; ++b) {
        const double x = dirs[3 * b], z = dirs[3 * b + 2];
        double y = dirs[3 * b + 1];
        if (x == 0 && y == 0) y += 1;                      /* zero_xy fix-up, ide_encoder.py:114-115 */
        const double kinv = roughness_ptr ? roughness_ptr[b] : roughness_scalar;
        double zp[17], re[17], im[17];
        zp[0] = 1; re[0] = 1; im[0] = 0;
        for (int k = 1; k <= lmax; ++k) {
            zp[k] = zp[k - 1] * z;
            re[k] = re[k - 1] * x - im[k - 1] * y; im[k] = re[k - 1] * y + im[k - 1] * x;
        }
        for (int i = 0; i < n; ++i) {
            double zc = 0;
            for (int k = 0; k <= lmax; ++k) zc += zp[k] * mat[(size_t)k * n + i];
            const float att = expf(-(0.5f * (float)(ml_l[i] * (ml_l[i] + 1))) * (float)kinv);   /* fp32 exp like torch */
            outputs[(size_t)b * 2 * n + i] = (float)(re[ml_m[i]] * zc) * att;
            outputs[(size_t)b * 2 * n + n + i] = (float)(im[ml_m[i]] * zc) * att;
        }
    }
    free(mat);
    return 0;
}

/* gradient of the encoding w.r.t. the direction and kappa_inv (torch autograd through ide_encoder.py:98-130): term
 * v = (x + i y)^m P(z) att; partials in closed form, in double, on the same fp32-rounded table as the forward */
API int oracle_ide_encode_backward(const float* grad, const float* dirs, const float* roughness_ptr, float roughness_scalar, uint32_t B,
                                   uint32_t deg_view, float* grad_dirs, float* grad_roughness) {
    if (deg_view < 1 || deg_view > 5) return -1;
    int ml_m[64], ml_l[64], n = 0;
    for (uint32_t i = 0; i < deg_view; ++i) { const int l = 1 << i; for (int m = 0; m <= l; ++m) { ml_m[n] = m; ml_l[n] = l; ++n; } }
    const int lmax = 1 << (deg_view - 1);
    double* mat = (double*)calloc((size_t)(lmax + 1) * n, sizeof(double));
    for (int i = 0; i < n; ++i)
        for (int k = 0; k <= ml_l[i] - ml_m[i]; ++k)
            mat[(size_t)k * n + i] = (double)(float)ide_coeff(ml_l[i], ml_m[i], k);
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)B; ++b) {
        const double x = dirs[3 * b], z = dirs[3 * b + 2];
        double y = dirs[3 * b + 1];
        if (x == 0 && y == 0) y += 1;
        const float kinv = roughness_ptr ? roughness_ptr[b] : roughness_scalar;
        double zp[17], re[17], im[17];
        zp[0] = 1; re[0] = 1; im[0] = 0;
        for (int k = 1; k <= lmax; ++k) {
            zp[k] = zp[k - 1] * z;
            re[k] = re[k - 1] * x - im[k - 1] * y; im[k] = re[k - 1] * y + im[k - 1] * x;
        }
        double gx = 0, gy = 0, gz = 0, gk = 0;
        for (int i = 0; i < n; ++i) {
            const int m = ml_m[i];
            double P = 0, dP = 0;
            for (int k = 0; k <= lmax; ++k) { P += zp[k] * mat[(size_t)k * n + i]; if (k) dP += k * zp[k - 1] * mat[(size_t)k * n + i]; }
            const float sigma = 0.5f * (float)(ml_l[i] * (ml_l[i] + 1));
            const double att = (double)expf(-sigma * kinv);
            const double a = grad[(size_t)b * 2 * n + i], c = grad[(size_t)b * 2 * n + n + i];
            if (m > 0) {
                gx += m * P * att * (a * re[m - 1] + c * im[m - 1]);
                gy += m * P * att * (c * re[m - 1] - a * im[m - 1]);
            }
            const double w = a * re[m] + c * im[m];
            gz += dP * att * w;
            gk -= (double)sigma * P * att * w;
        }
        if (grad_dirs) { grad_dirs[3 * b] = (float)gx; grad_dirs[3 * b + 1] = (float)gy; grad_dirs[3 * b + 2] = (float)gz; }
        if (grad_roughness) grad_roughness[b] = (float)gk;
    }
    free(mat);
    return 0;
}

