// raymarching operators for gfx950 -- C-ABI launchers + kernels.
//
// Replaces the reference extension raymarching/src/raymarching.cu (host functions :148-1054,
// declarations raymarching.h:7-18).  One lane per ray, 256-thread workgroups (one wave64 per
// SIMD); the marcher/compositor bodies live in march_core.hip.h so the fused render kernel
// runs the very same code.
#include "march_core.hip.h"

#include <map>
#include <mutex>
#include <utility>

#include <float.h>

using namespace envidr;

// ---------------------------------------------------------------------------------------------
// near/far from aabb (reference kernel_near_far_from_aabb, raymarching.cu:91-145)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void order2(float& a, float& b) {
    if (a > b) { const float c = a; a = b; b = c; }
}

__device__ __forceinline__ void near_far_of(const RayGeom& r, const float* __restrict__ aabb, const float min_near, float& out_near, float& out_far) {
    float near = (aabb[0] - r.ox) * r.rdx, far = (aabb[3] - r.ox) * r.rdx;
    order2(near, far);
    float ny = (aabb[1] - r.oy) * r.rdy, fy = (aabb[4] - r.oy) * r.rdy;
    order2(ny, fy);
    bool miss = near > fy || ny > far;
    if (!miss) {
        if (ny > near) near = ny;
        if (fy < far) far = fy;
        float nz = (aabb[2] - r.oz) * r.rdz, fz = (aabb[5] - r.oz) * r.rdz;
        order2(nz, fz);
        miss = near > fz || nz > far;
        if (!miss) {
            if (nz > near) near = nz;
            if (fz < far) far = fz;
            if (near < min_near) near = min_near;
        }
    }
    out_near = miss ? FLT_MAX : near;
    out_far = miss ? FLT_MAX : far;
}

__global__ void __launch_bounds__(kBlock) k_near_far_from_aabb(const float* __restrict__ rays_o,
                                                               const float* __restrict__ rays_d,
                                                               const float* __restrict__ aabb, uint32_t N,
                                                               float min_near, float* __restrict__ nears,
                                                               float* __restrict__ fars) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const RayGeom r = load_ray(rays_o, rays_d, n);
    near_far_of(r, aabb, min_near, nears[n], fars[n]);
}

// ---------------------------------------------------------------------------------------------
// background-sphere coordinates (reference kernel_sph_from_ray, raymarching.cu:162-198)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_sph_from_ray(const float* __restrict__ rays_o,
                                                         const float* __restrict__ rays_d, float radius,
                                                         uint32_t N, float* __restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const RayGeom r = load_ray(rays_o, rays_d, n);
    const float inv_pi = 0.3183098861837907f;

    // ||o + t d|| = radius, larger root (origin assumed inside the sphere)
    const float A = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz;
    const float Bh = r.ox * r.dx + r.oy * r.dy + r.oz * r.dz;
    const float Cc = r.ox * r.ox + r.oy * r.oy + r.oz * r.oz - radius * radius;
    const float t = (-Bh + sqrtf(Bh * Bh - A * Cc)) / A;

    const float x = r.ox + t * r.dx, y = r.oy + t * r.dy, z = r.oz + t * r.dz;
    // the reference calls the double overload of atan2 on float arguments
    const float theta = (float)atan2((double)sqrtf(x * x + z * z), (double)y);
    const float phi = (float)atan2((double)z, (double)x);
    coords[2 * (size_t)n] = 2 * theta * inv_pi - 1;
    coords[2 * (size_t)n + 1] = phi * inv_pi;
}

// ---------------------------------------------------------------------------------------------
// morton / packbits / scatter index (raymarching.cu:214-330)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_morton3D(const int32_t* __restrict__ coords, uint32_t N,
                                                     int32_t* __restrict__ indices) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int32_t* c = coords + (size_t)n * 3;
    indices[n] = (int32_t)morton_encode((uint32_t)c[0], (uint32_t)c[1], (uint32_t)c[2]);
}

__global__ void __launch_bounds__(kBlock) k_morton3D_invert(const int32_t* __restrict__ indices, uint32_t N,
                                                            int32_t* __restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int32_t code = indices[n];   // arithmetic shifts of the signed value, like the reference
    int32_t* c = coords + (size_t)n * 3;
    c[0] = (int32_t)compact3((uint32_t)(code >> 0));
    c[1] = (int32_t)compact3((uint32_t)(code >> 1));
    c[2] = (int32_t)compact3((uint32_t)(code >> 2));
}

// One lane packs one output byte from 8 consecutive floats = two 16-byte loads per lane.
__global__ void __launch_bounds__(kBlock) k_packbits(const float* __restrict__ grid, uint32_t N, float thresh,
                                                     uint8_t* __restrict__ bitfield) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float4* g = reinterpret_cast<const float4*>(grid + (size_t)n * 8);
    const float4 a = g[0], b = g[1];
    uint32_t bits = 0;
    bits |= (a.x > thresh) << 0; bits |= (a.y > thresh) << 1; bits |= (a.z > thresh) << 2; bits |= (a.w > thresh) << 3;
    bits |= (b.x > thresh) << 4; bits |= (b.y > thresh) << 5; bits |= (b.z > thresh) << 6; bits |= (b.w > thresh) << 7;
    bitfield[n] = (uint8_t)bits;
}

__global__ void __launch_bounds__(kBlock) k_get_scatter_idx(const int32_t* __restrict__ rays, uint32_t N,
                                                            int32_t* __restrict__ idx_map) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const uint32_t id = rays[3 * (size_t)n], off = rays[3 * (size_t)n + 1], cnt = rays[3 * (size_t)n + 2];
    for (uint32_t s = 0; s < cnt; ++s) idx_map[off + s] = (int32_t)id;
}

// ---------------------------------------------------------------------------------------------
// inference marcher (reference kernel_march_rays, raymarching.cu:839-944)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_march_rays(uint32_t n_alive, uint32_t n_step,
                                                       const int32_t* __restrict__ rays_alive,
                                                       const float* __restrict__ rays_t,
                                                       const float* __restrict__ rays_o,
                                                       const float* __restrict__ rays_d, MarchConsts k,
                                                       const float* __restrict__ fars, float* __restrict__ xyzs,
                                                       float* __restrict__ dirs, float* __restrict__ deltas,
                                                       const float* __restrict__ noises) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const uint32_t id = (uint32_t)rays_alive[n];
    const RayGeom r = load_ray(rays_o, rays_d, id);
    const float far = fars[id];

    float t = rays_t[id];
    float last_t = t;
    t += step_size(k, t) * noises[n];

    const size_t base = (size_t)n * n_step;
    float* px = xyzs + base * 3;
    float* pd = dirs + base * 3;
    float* pl = deltas + base * 2;
    // a FLAT loop, one cell visit per iteration (the statements of `for (s < n_step) if (!march_next(...)) break;`, per lane in the same
    // order): with the walk to the next sample nested inside the loop over samples a wave pays, at every sample index, the longest walk
    // of any of its lanes
    auto walk = [&](auto pow2) {
        uint32_t s = 0;
        while (s < n_step && t < far) {
            float x, y, z, dt;
            if (march_visit<decltype(pow2)::value>(k, r, t, x, y, z, dt)) {
                t += dt;
                px[0] = x; px[1] = y; px[2] = z;
                pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
                pl[0] = dt;
                pl[1] = t - last_t;
                last_t = t;
                px += 3; pd += 3; pl += 2;
                ++s;
            }
        }
    };
    if (k.H_pow2) walk(std::true_type{}); else walk(std::false_type{});
}

// ---------------------------------------------------------------------------------------------
// inference compositor (reference kernel_composite_rays, raymarching.cu:957-1046)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh,
                                                           uint32_t accum_deltas, uint32_t input_alpha,
                                                           int32_t* __restrict__ rays_alive,
                                                           float* __restrict__ rays_t,
                                                           const float* __restrict__ sigmas,
                                                           const float* __restrict__ rgbs,
                                                           const float* __restrict__ deltas,
                                                           float* __restrict__ weights_sum,
                                                           float* __restrict__ depth, float* __restrict__ image) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const uint32_t id = (uint32_t)rays_alive[n];
    const size_t base = (size_t)n * n_step;
    const float* ps = sigmas + base;
    const float* pc = rgbs + base * 3;
    const float* pl = deltas + base * 2;

    Accum a;
    a.t = rays_t[id];
    a.ws = weights_sum[id];
    a.depth = depth[id];
    a.r = image[3 * (size_t)id]; a.g = image[3 * (size_t)id + 1]; a.b = image[3 * (size_t)id + 2];

    uint32_t s = 0;
    for (; s < n_step; ++s) {
        const float d0 = pl[2 * s];
        if (d0 == 0) break;   // padded / exhausted sample
        const float alpha = alpha_from_sigma(ps[s], d0, input_alpha);
        if (composite_sample(a, alpha, pl[2 * s + 1], pc[3 * s], pc[3 * s + 1], pc[3 * s + 2], T_thresh,
                             accum_deltas))
            break;
    }
    if (s < n_step) rays_alive[n] = -1;
    else rays_t[id] = a.t;

    weights_sum[id] = a.ws;
    depth[id] = a.depth;
    image[3 * (size_t)id] = a.r; image[3 * (size_t)id + 1] = a.g; image[3 * (size_t)id + 2] = a.b;
}

// n_step = 4 or 8 (the reference's loop runs at n_step = min(N / n_alive, 8)): a ray's NS samples are 16 NS + 48 NS + 32 NS
// contiguous bytes, fetched as 16-byte loads up front instead of one dependent 4-byte load per sample and field at a 4 NS /
// 12 NS / 8 NS byte stride between lanes.  The recurrence is the kernel's above, statement for statement.
template <int NS>
__global__ void __launch_bounds__(kBlock) k_composite_rays_vec(uint32_t n_alive, float T_thresh, uint32_t accum_deltas, uint32_t input_alpha,
                                                               int32_t* __restrict__ rays_alive, float* __restrict__ rays_t,
                                                               const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                               const float* __restrict__ deltas, float* __restrict__ weights_sum,
                                                               float* __restrict__ depth, float* __restrict__ image) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const uint32_t id = (uint32_t)rays_alive[n];
    const size_t base = (size_t)n * NS;
    float ps[NS], pc[3 * NS], pl[2 * NS];
#pragma unroll
    for (int q = 0; q < NS / 4; ++q) {
        const float4 v = reinterpret_cast<const float4*>(sigmas + base)[q];
        ps[4 * q] = v.x; ps[4 * q + 1] = v.y; ps[4 * q + 2] = v.z; ps[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int q = 0; q < 3 * NS / 4; ++q) {
        const float4 v = reinterpret_cast<const float4*>(rgbs + base * 3)[q];
        pc[4 * q] = v.x; pc[4 * q + 1] = v.y; pc[4 * q + 2] = v.z; pc[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int q = 0; q < 2 * NS / 4; ++q) {
        const float4 v = reinterpret_cast<const float4*>(deltas + base * 2)[q];
        pl[4 * q] = v.x; pl[4 * q + 1] = v.y; pl[4 * q + 2] = v.z; pl[4 * q + 3] = v.w;
    }
    Accum a;
    a.t = rays_t[id];
    a.ws = weights_sum[id];
    a.depth = depth[id];
    a.r = image[3 * (size_t)id]; a.g = image[3 * (size_t)id + 1]; a.b = image[3 * (size_t)id + 2];
    bool done = false;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (!done) {
            const float d0 = pl[2 * s];
            if (d0 == 0) done = true;   // padded / exhausted sample
            else {
                const float alpha = alpha_from_sigma(ps[s], d0, input_alpha);
                done = composite_sample(a, alpha, pl[2 * s + 1], pc[3 * s], pc[3 * s + 1], pc[3 * s + 2], T_thresh, accum_deltas);
            }
        }
    }
    if (done) rays_alive[n] = -1;
    else rays_t[id] = a.t;
    weights_sum[id] = a.ws;
    depth[id] = a.depth;
    image[3 * (size_t)id] = a.r; image[3 * (size_t)id + 1] = a.g; image[3 * (size_t)id + 2] = a.b;
}

// ---------------------------------------------------------------------------------------------
// wave-level order-preserving compaction of the live ray ids (new; replaces the host-syncing
// boolean-mask gather of nerf/render_func/cuda_ray.py:345).
// Each 256-thread workgroup counts its survivors with wave ballots, reserves a contiguous output
// range in workgroup order through a decoupled look-back on per-block prefix slots, and scatters.
// Order preservation keeps results identical to the reference's rays_alive ordering.
// ---------------------------------------------------------------------------------------------
struct alignas(8) BlockPrefix { int32_t flag; int32_t value; };   // flag: 0 empty, 1 aggregate, 2 inclusive

__global__ void __launch_bounds__(kBlock) k_compact_alive(uint32_t n_alive, const int32_t* __restrict__ in_alive,
                                                          int32_t* __restrict__ out_alive,
                                                          int32_t* __restrict__ out_count,
                                                          unsigned long long* __restrict__ prefix_slots,
                                                          int32_t* __restrict__ ticket) {
    __shared__ int32_t s_wave_count[kBlock / 64];
    __shared__ int32_t s_block_base;
    __shared__ uint32_t s_block_id;

    // dynamic block id: dispatch order is not guaranteed, a ticket makes look-back deadlock free
    if (threadIdx.x == 0) s_block_id = (uint32_t)atomicAdd(ticket, 1);
    __syncthreads();
    const uint32_t bid = s_block_id;
    const uint32_t n = bid * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    const int32_t id = n < n_alive ? in_alive[n] : -1;
    const bool keep = id >= 0;
    const unsigned long long mask = __ballot(keep);
    const int32_t rank_in_wave = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave_count[wave] = __popcll(mask);
    __syncthreads();

    int32_t wave_base = 0, block_total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kBlock / 64; ++w) {
        const int32_t c = s_wave_count[w];
        if (w < wave) wave_base += c;
        block_total += c;
    }

    if (threadIdx.x == 0) {
        // publish aggregate, then look back over predecessors (single-word {flag,value} granules)
        auto pack = [](int32_t flag, int32_t value) {
            return ((unsigned long long)(uint32_t)flag << 32) | (uint32_t)value;
        };
        if (bid == 0) {
            __hip_atomic_store(&prefix_slots[0], pack(2, block_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_block_base = 0;
        } else {
            __hip_atomic_store(&prefix_slots[bid], pack(1, block_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int32_t running = 0;
            int32_t look = (int32_t)bid - 1;
            while (true) {
                const unsigned long long v =
                    __hip_atomic_load(&prefix_slots[look], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int32_t flag = (int32_t)(v >> 32);
                if (flag == 0) { __builtin_amdgcn_s_sleep(1); continue; }
                running += (int32_t)(uint32_t)v;
                if (flag == 2) break;
                --look;
            }
            __hip_atomic_store(&prefix_slots[bid], pack(2, running + block_total), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            s_block_base = running;
        }
        if ((bid + 1) * blockDim.x >= n_alive) *out_count = s_block_base + block_total;
    }
    __syncthreads();
    if (keep) out_alive[s_block_base + wave_base + rank_in_wave] = id;
}

// ---------------------------------------------------------------------------------------------
// training marcher (reference kernel_march_rays_train, raymarching.cu:340-508)
// ---------------------------------------------------------------------------------------------
template <int BLOCK, int kTimeCache>
__global__ void __launch_bounds__(BLOCK) k_march_rays_train(const float* __restrict__ rays_o,
                                                             const float* __restrict__ rays_d, MarchConsts k,
                                                             uint32_t early_stop_steps, uint32_t N, uint32_t M,
                                                             const float* __restrict__ nears,
                                                             const float* __restrict__ fars,
                                                             float* __restrict__ xyzs, float* __restrict__ dirs,
                                                             float* __restrict__ deltas, int32_t* __restrict__ rays,
                                                             int32_t* __restrict__ counter,
                                                             const float* __restrict__ noises) {
    // The reference's second pass walks every ray again.  The ray times of the first kTimeCache samples found by the counting pass stay
    // in LDS ([sample][thread]: conflict-free); position and step are pure functions of that time (march_visit), so the write pass
    // replays them -- same statements, same bits -- and only marches on from where the cache ends.
    __shared__ float s_time[kTimeCache][BLOCK];
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const RayGeom r = load_ray(rays_o, rays_d, n);
    const float near = nears[n], far = fars[n];

    float t0 = near;
    t0 += step_size(k, t0) * noises[n];

    // pass 1: count occupied steps
    float t = t0, x, y, z, dt;
    uint32_t num_steps = 0;
    // (a FLAT loop, one cell visit per iteration: with `while (march_next(...))` -- a walk to the next sample inside a loop over samples
    //  -- a wave pays, at every sample index, the longest walk any of its lanes takes there; the visits are the same statements in
    //  the same order for every lane, march_core.hip.h)
    auto count = [&](auto pow2) {
        while (t < far && num_steps < early_stop_steps) {
            if (march_visit<decltype(pow2)::value>(k, r, t, x, y, z, dt)) {
                if (num_steps < kTimeCache) s_time[num_steps][threadIdx.x] = t;
                ++num_steps;
                t += dt;
            }
        }
    };
    if (k.H_pow2) count(std::true_type{}); else count(std::false_type{});

    // reserve output ranges (the compiler folds these into one atomic per wave)
    const uint32_t point_index = (uint32_t)atomicAdd(counter, (int32_t)num_steps);
    const uint32_t ray_index = (uint32_t)atomicAdd(counter + 1, 1);
    rays[3 * (size_t)ray_index] = (int32_t)n;
    rays[3 * (size_t)ray_index + 1] = (int32_t)point_index;
    rays[3 * (size_t)ray_index + 2] = (int32_t)num_steps;
    if (num_steps == 0 || point_index + num_steps > M) return;

    // pass 2: re-march and write
    float* px = xyzs + (size_t)point_index * 3;
    float* pd = dirs + (size_t)point_index * 3;
    float* pl = deltas + (size_t)point_index * 2;
    float last_t = near;
    auto emit = [&]() {
        px[0] = x; px[1] = y; px[2] = z;
        pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
        pl[0] = dt;
        pl[1] = t - last_t;
        last_t = t;
        px += 3; pd += 3; pl += 2;
    };
    const uint32_t cached = min(num_steps, (uint32_t)kTimeCache);
    t = t0;
    for (uint32_t s = 0; s < cached; ++s) {
        const float ts = s_time[s][threadIdx.x];
        x = clampf(r.ox + ts * r.dx, -k.bound, k.bound);          // the statements of march_visit at the sample's time
        y = clampf(r.oy + ts * r.dy, -k.bound, k.bound);
        z = clampf(r.oz + ts * r.dz, -k.bound, k.bound);
        dt = step_size(k, ts);
        t = ts + dt;
        emit();
    }
    // past the cache: march on, again one cell visit per iteration
    uint32_t s = cached;
    auto rest = [&](auto pow2) {
        while (t < far && s < num_steps) {
            if (march_visit<decltype(pow2)::value>(k, r, t, x, y, z, dt)) {
                t += dt;
                emit();
                ++s;
            }
        }
    };
    if (s < num_steps) {
        if (k.H_pow2) rest(std::true_type{}); else rest(std::false_type{});
    }
}

// ---------------------------------------------------------------------------------------------
// training compositor forward / backward (raymarching.cu:529-832)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_composite_train_fwd(const float* __restrict__ sigmas,
                                                                const float* __restrict__ rgbs,
                                                                const float* __restrict__ deltas,
                                                                const int32_t* __restrict__ rays, uint32_t M,
                                                                uint32_t N, float T_thresh, uint32_t accum_deltas,
                                                                uint32_t input_alpha, float* __restrict__ weights_sum,
                                                                float* __restrict__ depth, float* __restrict__ image,
                                                                float* __restrict__ weights) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const uint32_t id = rays[3 * (size_t)n], off = rays[3 * (size_t)n + 1], cnt = rays[3 * (size_t)n + 2];

    float r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
    if (cnt != 0 && off + cnt <= M) {
        float T = 1.0f;
        for (uint32_t s = 0; s < cnt; ++s) {
            const size_t m = (size_t)off + s;
            const float alpha = alpha_from_sigma(sigmas[m], deltas[2 * m], input_alpha);
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
    weights_sum[id] = ws;
    depth[id] = d;
    image[3 * (size_t)id] = r; image[3 * (size_t)id + 1] = g; image[3 * (size_t)id + 2] = b;
}

__global__ void __launch_bounds__(kBlock) k_composite_train_bwd(
    const float* __restrict__ grad_weights_sum, const float* __restrict__ grad_image,
    const float* __restrict__ grad_depth, const float* __restrict__ sigmas, const float* __restrict__ rgbs,
    const float* __restrict__ deltas, const int32_t* __restrict__ rays, const float* __restrict__ weights_sum,
    const float* __restrict__ image, const float* __restrict__ depth, uint32_t M, uint32_t N, float T_thresh,
    float* __restrict__ grad_sigmas, float* __restrict__ grad_rgbs, uint32_t accum_deltas, uint32_t input_alpha) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const uint32_t id = rays[3 * (size_t)n], off = rays[3 * (size_t)n + 1], cnt = rays[3 * (size_t)n + 2];
    if (cnt == 0 || off + cnt > M) return;

    const float gr = grad_image[3 * (size_t)id], gg = grad_image[3 * (size_t)id + 1], gb = grad_image[3 * (size_t)id + 2];
    const float gws = grad_weights_sum[id];
    const float r_final = image[3 * (size_t)id], g_final = image[3 * (size_t)id + 1], b_final = image[3 * (size_t)id + 2];
    const float ws_final = weights_sum[id];
    // Reference quirk kept on purpose (raymarching.cu:771,797): the depth terms read element 0 of
    // `depth` / `grad_depth`, not the ray's own element -- those two pointers are never offset.
    const float d_final = depth[0], gd = grad_depth[0];

    float T = 1.0f, r = 0, g = 0, b = 0, t = 0, d = 0;
    for (uint32_t s = 0; s < cnt; ++s) {
        const size_t m = (size_t)off + s;
        const float alpha = alpha_from_sigma(sigmas[m], deltas[2 * m], input_alpha);
        const float w = alpha * T;
        const float grad_scale = input_alpha ? (1.0f / (1.0f - alpha + 1e-4f)) : deltas[2 * m];
        const float cr = rgbs[3 * m], cg = rgbs[3 * m + 1], cb = rgbs[3 * m + 2];
        r += w * cr; g += w * cg; b += w * cb;
        t = accum_deltas ? t + deltas[2 * m + 1] : deltas[2 * m + 1];
        d += w * t;
        T *= 1.0f - alpha;

        grad_rgbs[3 * m] = gr * w; grad_rgbs[3 * m + 1] = gg * w; grad_rgbs[3 * m + 2] = gb * w;
        grad_sigmas[m] = grad_scale * (gr * (T * cr - (r_final - r)) + gg * (T * cg - (g_final - g)) +
                                       gb * (T * cb - (b_final - b)) + gd * (T * t - (d_final - d)) +
                                       gws * (1 - ws_final));
        if (T < T_thresh) break;
    }
}

// ---- ray generation (reference nerf/utils.py:109-209, full-image branch :193-207) -----------------------------------------
// One lane per (camera, pixel): pixel centre (+0.5), camera-space direction ((i - cx) / fx, (j - cy) / fy, 1) normalised, rotated by
// the camera-to-world pose; the origin is the pose's translation.  fp32 throughout, the reference's operation order.
__global__ void __launch_bounds__(kBlock) k_get_rays(const float* __restrict__ poses, float fx, float fy, float cx, float cy, uint32_t H,
                                                     uint32_t W, uint32_t B, float* __restrict__ rays_o, float* __restrict__ rays_d) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per = (size_t)H * W;
    if (id >= per * B) return;
    const uint32_t b = (uint32_t)(id / per), pix = (uint32_t)(id % per);
    const float i = (float)(pix % W) + 0.5f, j = (float)(pix / W) + 0.5f;
    const float xs = (i - cx) / fx * 1.0f, ys = (j - cy) / fy * 1.0f, zs = 1.0f;
    const float inv = sqrtf(xs * xs + ys * ys + zs * zs);
    const float dx = xs / inv, dy = ys / inv, dz = zs / inv;
    const float* P = poses + 16 * (size_t)b;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rays_d[3 * id + k] = dx * P[4 * k] + dy * P[4 * k + 1] + dz * P[4 * k + 2];
        rays_o[3 * id + k] = P[4 * k + 3];
    }
}

// =============================================================================================
// C-ABI launchers
// =============================================================================================
#define LAUNCH_1D(kernel, count, stream, ...)                                                            \
    do {                                                                                                 \
        if ((count) == 0) return ENVIDR_OK;                                                              \
        hipLaunchKernelGGL(kernel, dim3(ceil_div((count), kBlock)), dim3(kBlock), 0, as_stream(stream),  \
                           __VA_ARGS__);                                                                 \
        return check_launch(#kernel);                                                                    \
    } while (0)

extern "C" {

int envidr_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                              float min_near, float* nears, float* fars, envidr_stream_t stream) {
    ENVIDR_REQUIRE(N == 0 || (rays_o && rays_d && aabb && nears && fars), "near_far_from_aabb: null pointer");
    // (four rays per lane with 16-byte accesses was measured slower at 640 k rays -- 22 us against 12: a quarter of the waves, each
    //  with one dependent load -> compute -> store chain, leaves fewer bytes in flight than one ray per lane; the call is 20 MB)
    LAUNCH_1D(k_near_far_from_aabb, N, stream, rays_o, rays_d, aabb, N, min_near, nears, fars);
}

int envidr_get_rays(const float* poses, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W, uint32_t B, float* rays_o,
                    float* rays_d, envidr_stream_t stream) {
    ENVIDR_REQUIRE(B == 0 || H == 0 || W == 0 || (poses && rays_o && rays_d), "get_rays: null pointer");
    const unsigned long long n = (unsigned long long)H * W * B;
    ENVIDR_REQUIRE(n < (1ull << 32), "get_rays: too many rays for one call");
    LAUNCH_1D(k_get_rays, (uint32_t)n, stream, poses, fx, fy, cx, cy, H, W, B, rays_o, rays_d);
}

int envidr_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords,
                        envidr_stream_t stream) {
    ENVIDR_REQUIRE(N == 0 || (rays_o && rays_d && coords), "sph_from_ray: null pointer");
    LAUNCH_1D(k_sph_from_ray, N, stream, rays_o, rays_d, radius, N, coords);
}

int envidr_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, envidr_stream_t stream) {
    ENVIDR_REQUIRE(N == 0 || (coords && indices), "morton3D: null pointer");
    LAUNCH_1D(k_morton3D, N, stream, coords, N, indices);
}

int envidr_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, envidr_stream_t stream) {
    ENVIDR_REQUIRE(N == 0 || (coords && indices), "morton3D_invert: null pointer");
    LAUNCH_1D(k_morton3D_invert, N, stream, indices, N, coords);
}

int envidr_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield,
                    envidr_stream_t stream) {
    ENVIDR_REQUIRE(N == 0 || (grid && bitfield), "packbits: null pointer");
    ENVIDR_REQUIRE((reinterpret_cast<uintptr_t>(grid) & 15) == 0, "packbits: grid must be 16-byte aligned");
    LAUNCH_1D(k_packbits, N, stream, grid, N, density_thresh, bitfield);
}

int envidr_get_scatter_idx(const int32_t* rays, uint32_t N, int32_t* idx_map, envidr_stream_t stream) {
    ENVIDR_REQUIRE(N == 0 || (rays && idx_map), "get_scatter_idx: null pointer");
    LAUNCH_1D(k_get_scatter_idx, N, stream, rays, N, idx_map);
}

int envidr_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                      const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                      uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars,
                      float* xyzs, float* dirs, float* deltas, const float* noises, envidr_stream_t stream) {
    (void)nears;   // the reference kernel loads nears[index] but never uses it
    ENVIDR_REQUIRE(n_alive == 0 || (rays_alive && rays_t && rays_o && rays_d && grid && fars && xyzs && dirs &&
                                    deltas && noises),
                   "march_rays: null pointer");
    ENVIDR_REQUIRE(C >= 1 && H >= 1 && max_steps >= 1, "march_rays: C, H, max_steps must be >= 1");
    const MarchConsts k = make_march_consts(bound, dt_gamma, max_steps, C, H, grid);
    LAUNCH_1D(k_march_rays, n_alive, stream, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, k, fars, xyzs,
              dirs, deltas, noises);
}

int envidr_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, uint32_t accum_deltas,
                          uint32_t input_alpha, int32_t* rays_alive, float* rays_t, const float* sigmas,
                          const float* rgbs, const float* deltas, float* weights_sum, float* depth, float* image,
                          envidr_stream_t stream) {
    ENVIDR_REQUIRE(n_alive == 0 || (rays_alive && rays_t && sigmas && rgbs && deltas && weights_sum && depth && image),
                   "composite_rays: null pointer");
    const bool al = ((reinterpret_cast<uintptr_t>(sigmas) | reinterpret_cast<uintptr_t>(rgbs) | reinterpret_cast<uintptr_t>(deltas)) & 15) == 0;
    if (al && n_step == 8) {
        LAUNCH_1D(k_composite_rays_vec<8>, n_alive, stream, n_alive, T_thresh, accum_deltas, input_alpha, rays_alive, rays_t, sigmas, rgbs, deltas,
                  weights_sum, depth, image);
    }
    if (al && n_step == 4) {
        LAUNCH_1D(k_composite_rays_vec<4>, n_alive, stream, n_alive, T_thresh, accum_deltas, input_alpha, rays_alive, rays_t, sigmas, rgbs, deltas,
                  weights_sum, depth, image);
    }
    LAUNCH_1D(k_composite_rays, n_alive, stream, n_alive, n_step, T_thresh, accum_deltas, input_alpha, rays_alive,
              rays_t, sigmas, rgbs, deltas, weights_sum, depth, image);
}

int envidr_compact_alive(uint32_t n_alive, const int32_t* rays_alive, int32_t* out_alive, int32_t* out_count,
                         envidr_stream_t stream) {
    ENVIDR_REQUIRE(out_count != nullptr, "compact_alive: out_count is null");
    hipStream_t s = as_stream(stream);
    if (n_alive == 0) {
        if (hipMemsetAsync(out_count, 0, sizeof(int32_t), s) != hipSuccess) return check_launch("compact_alive memset");
        return ENVIDR_OK;
    }
    ENVIDR_REQUIRE(rays_alive && out_alive, "compact_alive: null pointer");
    // look-back scratch: one 8-byte slot per workgroup + a ticket.  One buffer per (device, stream), grown on demand and
    // kept: launches on different streams or devices never share ticket / prefix slots (the kernel spins on them).
    struct Scratch { void* ptr; size_t bytes; };
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Scratch> pool;
    const uint32_t blocks = ceil_div(n_alive, kBlock);
    const size_t need = (size_t)blocks * sizeof(unsigned long long) + 16;
    void* scratch = nullptr;
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lock(mu);
        Scratch& sc = pool[{dev, s}];
        if (need > sc.bytes) {
            if (sc.ptr) (void)hipFree(sc.ptr);
            sc.bytes = need * 2;
            if (hipMalloc(&sc.ptr, sc.bytes) != hipSuccess) {
                sc.ptr = nullptr; sc.bytes = 0;
                return check_launch("compact_alive scratch alloc");
            }
        }
        scratch = sc.ptr;
    }
    if (hipMemsetAsync(scratch, 0, need, s) != hipSuccess) return check_launch("compact_alive memset");
    int32_t* ticket = reinterpret_cast<int32_t*>(scratch);
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(scratch) + 16);
    hipLaunchKernelGGL(k_compact_alive, dim3(blocks), dim3(kBlock), 0, s, n_alive, rays_alive, out_alive, out_count,
                       slots, ticket);
    return check_launch("k_compact_alive");
}

int envidr_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                            float dt_gamma, uint32_t max_steps, uint32_t early_stop_steps, uint32_t N, uint32_t C,
                            uint32_t H, uint32_t M, const float* nears, const float* fars, float* xyzs, float* dirs,
                            float* deltas, int32_t* rays, int32_t* counter, const float* noises,
                            envidr_stream_t stream) {
    ENVIDR_REQUIRE(N == 0 || (rays_o && rays_d && grid && nears && fars && xyzs && dirs && deltas && rays && counter &&
                              noises),
                   "march_rays_train: null pointer");
    ENVIDR_REQUIRE(C >= 1 && H >= 1 && max_steps >= 1, "march_rays_train: C, H, max_steps must be >= 1");
    const MarchConsts k = make_march_consts(bound, dt_gamma, max_steps, C, H, grid);
    if (N == 0) return ENVIDR_OK;
    // A training batch is a few thousand rays: the run time is the latency of the longest walk, so one wave per workgroup (every wave on
    // its own CU) and a cache that holds nearly every ray's samples; a frame-sized call is throughput bound and keeps the LDS footprint
    // per wave small.  Measured, profiles/r05l/march_train_probe.txt.
#define MARCH_TRAIN(B, T) hipLaunchKernelGGL((k_march_rays_train<B, T>), dim3(ceil_div(N, (uint32_t)B)), dim3(B), 0, as_stream(stream), \
                                              rays_o, rays_d, k, early_stop_steps, N, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises)
    if (N <= 32768u) MARCH_TRAIN(64, 192);
    else MARCH_TRAIN(256, 48);
#undef MARCH_TRAIN
    return check_launch("k_march_rays_train");
}

int envidr_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas,
                                        const int32_t* rays, uint32_t M, uint32_t N, float T_thresh,
                                        uint32_t accum_deltas, uint32_t input_alpha, float* weights_sum,
                                        float* depth, float* image, float* weights, envidr_stream_t stream) {
    ENVIDR_REQUIRE(N == 0 || (rays && weights_sum && depth && image), "composite_rays_train_forward: null pointer");
    ENVIDR_REQUIRE(N == 0 || M == 0 || (sigmas && rgbs && deltas), "composite_rays_train_forward: null sample array");
    LAUNCH_1D(k_composite_train_fwd, N, stream, sigmas, rgbs, deltas, rays, M, N, T_thresh, accum_deltas,
              input_alpha, weights_sum, depth, image, weights);
}

int envidr_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image,
                                         const float* grad_depth, const float* sigmas, const float* rgbs,
                                         const float* deltas, const int32_t* rays, const float* weights_sum,
                                         const float* image, const float* depth, uint32_t M, uint32_t N,
                                         float T_thresh, float* grad_sigmas, float* grad_rgbs, uint32_t accum_deltas,
                                         uint32_t input_alpha, envidr_stream_t stream) {
    ENVIDR_REQUIRE(N == 0 || (grad_weights_sum && grad_image && grad_depth && sigmas && rgbs && deltas && rays &&
                              weights_sum && image && depth && grad_sigmas && grad_rgbs),
                   "composite_rays_train_backward: null pointer");
    LAUNCH_1D(k_composite_train_bwd, N, stream, grad_weights_sum, grad_image, grad_depth, sigmas, rgbs, deltas, rays,
              weights_sum, image, depth, M, N, T_thresh, grad_sigmas, grad_rgbs, accum_deltas, input_alpha);
}

}  // extern "C"
