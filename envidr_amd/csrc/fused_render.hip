// Fused persistent-wave inference renderer for gfx950 (include/envidr_render.h).
//
// One launch renders a whole ray batch.  Every wave64 is a persistent worker that owns 64 ray
// slots (one per lane) and loops:
//
//   refill   lanes without a ray pull the next ray id from a global queue head (one aggregated
//            atomic per wave), intersect it with the scene box and start marching;
//   march    each lane advances its ray to the next OCCUPIED sample (march_core.hip.h -- the very
//            code of the standalone march_rays operator, bit-exact with the reference);
//   shade    the wave shades its 64 samples: per-lane hash-grid gathers with analytic Jacobian,
//            then the MLPs on the matrix cores with activations kept in registers
//            (mlp_mfma.hip.h): SDF forward + input-gradient (normals), 2 x IDE, 2 x environment
//            MLP, diffuse and specular heads;
//   blend    each lane composites its sample into its ray's accumulators; finished rays are
//            written out and the lane becomes free again.
//
// What this removes relative to the reference loop (nerf/render_func/cuda_ray.py:277-346): every
// per-iteration allocation / memset, the [M,*] sample tensors round-tripping HBM, ~40 small kernel
// launches + 16 cuBLAS GEMMs per iteration, the autograd graph (incl. a 48.8 MB table-gradient
// memset + scatter per iteration), the boolean-mask compaction and its host sync, and all shading
// of padded / already-terminated samples: a lane only ever shades a sample its ray will composite.
//
// Equivalence to the reference: a ray's samples, step sizes and occupancy decisions are those of
// the reference marcher resumed from the composited ray time after every sample, i.e. the
// reference loop with n_step = 1; its per-ray compositing recurrence is the reference's.  The
// reference's larger n_step only changes where the fp32 ray time is re-derived from the summed
// deltas (ulp-level), see DESIGN.md.
#include "fused_common.hip.h"
#include "sh_core.hip.h"
#include "rowio.hip.h"

#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

using namespace envidr;

namespace {

constexpr uint32_t kMaxGroup = 8;   // largest number of lanes (consecutive samples) per ray in tail mode

struct RenderArgs {
    const float* rays_o;
    const float* rays_d;
    uint32_t N;
    MarchConsts mk;
    Aabb box;
    float min_near, T_thresh, density_scale, bg;
    uint32_t max_samples;
    // hash grid
    const float* table;
    HashLevelK lv[kLevels];
    uint32_t num_levels;
    float bound2;                 // 2 * bound
    // weights in consumption order, one blob per pass (see envidr_render.h), padded to whole 16 KiB chunks
    const float* sdf_blob;
    const float* env_blob;
    const float* head_blob;
    const float* sdf_w3r0;      // row 0 of the last SDF layer as a packed row vector
    float inv_beta, beta;
    float rough_bias, rough_act_scale, rough_scale;
    // env / heads
    float kappa_diffuse, light_scale, intensity_scale;
    int has_rot;
    float rot[9];
    // pass selection (indirect-reflection rendering, renderer.py:437-513)
    int geometry_only;           // composite normals only: no shading passes at all
    const float* r_images;       // [N,4] per-ray reflected radiance (rgb, visibility) or null
    const float* renv_blob;      // R1..R4 of the reflected-radiance feature MLP 4 -> 64 -> 64 -> 64 -> 12
    const float* spec2_blob;     // S1..S3: the specular head once more, for the renv branch
    float indir_rough_thresh;
    // per-sample geometry export (geometry cache for re-lighting / env rotation; only with geometry_only)
    uint32_t* ex_counter;        // [1] records appended (keeps counting past ex_capacity: the caller sizes a retry from it)
    uint32_t ex_capacity;
    uint32_t* ex_ray; uint32_t* ex_idx;      // [cap] ray id, index of the sample within its ray
    float* ex_w; float* ex_normal; float* ex_geo; float* ex_rough;   // [cap] compositing weight, [cap,3], [cap,12], [cap]
    // outputs
    float* image; float* depth; float* ws; float* normal; float* diffuse; float* specular; float* roughness;
    unsigned long long* stats;
    uint32_t* ray_counter;      // [0] queue head, [1] number of hit rays (written by k_first_hit)
    uint16_t* ray_cost;         // [N] optional scheduling hint, in: samples each ray took in an earlier render, out: this render's
    const uint32_t* hit_ids;    // [N] compacted ids of rays that have at least one sample
    const float* hit_t;         // [N] per RAY: marcher time at its first sample
};

// cost buckets of the optional scheduling hint: 256 buckets of 2^kCostShift samples (coarse buckets keep image-space
// neighbours, whose hash gathers share cache lines, together)
constexpr uint32_t kCostShift = 4;
constexpr uint32_t kCostBuckets = 256, kCostBase = 4;      // counters: [0] queue head, [1] hits, [4..] bucket counts, then bucket cursors
constexpr uint32_t kScratchCounterWords = kCostBase + 2 * kCostBuckets;
__device__ __forceinline__ uint32_t cost_bucket(uint32_t cost) { return min(cost >> kCostShift, kCostBuckets - 1); }

// First-hit pre-pass: one lane per ray, full occupancy.  Empty-space skipping is cheap per ray but
// long and divergent; inside the persistent kernel it would stall 63 shading lanes behind one
// marching lane.  Rays with no occupied sample are finished here (background only); the others are
// appended to a compact work list, wave by wave so neighbouring rays stay together.
__global__ void __launch_bounds__(kBlock) k_first_hit(const RenderArgs a, uint32_t* __restrict__ hit_ids,
                                                      float* __restrict__ hit_t, uint32_t* __restrict__ counters) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    if (id < a.N) {
        const RayGeom rg = load_ray(a.rays_o, a.rays_d, id);
        float near, far, x, y, z, dt, t_at = 0;
        near_far(rg, a.box, a.min_near, near, far);
        float t = near;
        hit = march_next(a.mk, rg, far, t, x, y, z, dt, &t_at);
        if (hit) {
            hit_t[id] = t_at;
        } else {
            a.image[3 * (size_t)id] = a.bg; a.image[3 * (size_t)id + 1] = a.bg; a.image[3 * (size_t)id + 2] = a.bg;
            a.depth[id] = 0;
            a.ws[id] = 0;
            if (a.normal) { a.normal[3 * (size_t)id] = 0; a.normal[3 * (size_t)id + 1] = 0; a.normal[3 * (size_t)id + 2] = 0; }
            if (a.diffuse) { a.diffuse[3 * (size_t)id] = 0; a.diffuse[3 * (size_t)id + 1] = 0; a.diffuse[3 * (size_t)id + 2] = 0; }
            if (a.specular) { a.specular[3 * (size_t)id] = 0; a.specular[3 * (size_t)id + 1] = 0; a.specular[3 * (size_t)id + 2] = 0; }
            if (a.roughness) a.roughness[id] = 0;
        }
    }
    const unsigned long long mask = __ballot(hit);
    const uint32_t lane = threadIdx.x & 63;
    if (a.ray_cost) {
        // cost-ordered work list: count the hit rays per cost bucket here, k_order_hits places them
        if (id < a.N) { if (!hit) hit_t[id] = -1.0f; }
        const uint32_t bucket = hit ? cost_bucket(a.ray_cost[id]) : 0u;
        unsigned long long todo = mask;
        while (todo) {                                       // one atomic per distinct bucket in the wave
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t b = __shfl(bucket, leader);
            const unsigned long long same = __ballot(hit && bucket == b);
            if ((int)lane == leader) atomicAdd(&counters[kCostBase + b], (uint32_t)__popcll(same));
            todo &= ~same;
        }
        if (lane == 0 && mask) atomicAdd(&counters[1], (uint32_t)__popcll(mask));
        return;
    }
    uint32_t base = 0;
    if (lane == 0 && mask) base = atomicAdd(&counters[1], (uint32_t)__popcll(mask));
    base = __shfl(base, 0);
    if (hit) hit_ids[base + __popcll(mask & ((1ull << lane) - 1ull))] = id;
}

// Second half of the cost-ordered work list (longest rays first: the persistent waves then run out of work together
// instead of a few waves finishing long rays alone; a poor hint only costs balance, never correctness).  Bucket b's
// rays go to [start_b, start_b + count_b) with start_b = number of hit rays in more expensive buckets; inside a bucket
// the rays of one wave stay together.
__global__ void __launch_bounds__(kBlock) k_order_hits(const RenderArgs a, uint32_t* __restrict__ hit_ids, const float* __restrict__ hit_t,
                                                       uint32_t* __restrict__ counters) {
    __shared__ uint32_t start[kCostBuckets];
    if (threadIdx.x < 64) {
        // exclusive suffix sum over the bucket counts, by one wave
        uint32_t run = 0;
        for (int hi = kCostBuckets - 64; hi >= 0; hi -= 64) {
            const int b = hi + 63 - (int)threadIdx.x;                 // lane 0 takes the most expensive bucket of the group
            const uint32_t c = counters[kCostBase + b];
            uint32_t incl = c;
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off);
                if ((int)threadIdx.x >= off) incl += up;
            }
            start[b] = run + incl - c;
            run += __shfl(incl, 63);
        }
    }
    __syncthreads();
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    const bool hit = id < a.N && hit_t[id] >= 0.0f;
    const uint32_t bucket = hit ? cost_bucket(a.ray_cost[id]) : 0u;
    if (id < a.N && !hit) a.ray_cost[id] = 0;
    unsigned long long todo = __ballot(hit);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t b = __shfl(bucket, leader);
        const unsigned long long same = __ballot(hit && bucket == b);
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(&counters[kCostBase + kCostBuckets + b], (uint32_t)__popcll(same));
        base = __shfl(base, leader);
        if (hit && bucket == b) hit_ids[start[b] + base + __popcll(same & ((1ull << lane) - 1ull))] = id;
        todo &= ~same;
    }
}

// heads: diffuse DSTEPS lane steps -> 32 -> 3, specular SSTEPS lane steps -> 64 -> 64 -> 3 (a lane step = 2 input features)
template <int DSTEPS, int SSTEPS>
struct HeadLayout {
    static constexpr int D1 = 0, D2 = D1 + lane_layer_frags(DSTEPS, 1, true), S1 = D2 + tile_layer_frags(1, 1, true),
                         S2 = S1 + lane_layer_frags(SSTEPS, 2, true), S3 = S2 + tile_layer_frags(2, 2, true),
                         Frags = S3 + tile_layer_frags(2, 1, true);
};

// Weight delivery: every wave streams the pass blobs from L2 through its own register ring (WeightRing, mlp_mfma.hip.h),
// single-wave workgroups.  (A 4-wave LDS-shared stream was measured in round 1 and was slower at the fp32 MFMA rate -- its
// per-chunk barriers cost more than the 4x L2 traffic it saves)
constexpr int kRingDepth = 32;
constexpr int kHashAhead = 2;         // hash levels whose corner gathers are in flight ahead of the one being interpolated
constexpr uint32_t kBlockThreads = 64;
constexpr int ring_padded(int frags) { return (frags + kRingDepth - 1) / kRingDepth * kRingDepth; }

// reflected-radiance branch (network.py:612-659): renv MLP 4 -> 64 -> 64 -> 64 -> 12, then the specular head again
constexpr int kRenv1 = 0, kRenv2 = kRenv1 + lane_layer_frags(2, 2, true), kRenv3 = kRenv2 + tile_layer_frags(2, 2, true),
              kRenv4 = kRenv3 + tile_layer_frags(2, 2, true), kRenvFrags = kRenv4 + tile_layer_frags(2, 1, true);
constexpr int kSpec2S1 = 0, kSpec2S2 = kSpec2S1 + lane_layer_frags(14, 2, true), kSpec2S3 = kSpec2S2 + tile_layer_frags(2, 2, true),
              kSpec2Frags = kSpec2S3 + tile_layer_frags(2, 1, true);

// ---- shading of one sample per lane: environment MLP twice (IDE of the rotated normal and of the reflected
// direction), then the diffuse and specular heads (network.py:524-698).  Shared by the persistent render kernel and
// the shade-only kernel.  The weight source `wp` must already be streaming env_blob (env family) or head_blob
// (no-env family) when this is entered; on return it is streaming c.next_blob.
struct ShadeConsts {
    const float* env_blob;
    const float* head_blob;
    const float* next_blob;      // the blob the caller's next pass consumes
    uint32_t next_chunks;
    float kappa_diffuse, light_scale;
};

// PRE_ENV: the environment features were computed beforehand (split-precision mode, shade_split.hip): env_pre points at this
// lane's 24 floats env(normal) | env(reflection) and `wp` streams head_blob on entry.
template <int IDE_DEG, int ENV_T, int SH_DEG, bool PRE_ENV = false, class WP, class Tick>
__device__ __forceinline__ void shade_sample(WP& wp, const uint32_t lane, const ShadeConsts c, const float (&nrm)[3],
                                             const float (&nenv)[3], const float (&wr)[3], const float (&vd)[3], const float ndot,
                                             const float (&geo)[12], const float rough, float (&cd)[3], float (&cs)[3], float (&env_r)[12], Tick&& tick,
                                             const EnvAux& aux, const float* env_pre = nullptr) {
    constexpr bool kEnvNet = SH_DEG == 0;
    constexpr int kShDim = SH_DEG * SH_DEG;
    constexpr int kDiffIn = kEnvNet ? 24 : 12, kSpecIn = kEnvNet ? 28 : 2 * kShDim + 13;
    constexpr int kDSteps = (kDiffIn + 1) / 2, kSSteps = (kSpecIn + 1) / 2;
    using Head = HeadLayout<kDSteps, kSSteps>;
    constexpr int kHeadD1 = Head::D1, kHeadD2 = Head::D2, kHeadS1 = Head::S1, kHeadS2 = Head::S2, kHeadS3 = Head::S3,
                  kHeadFrags = Head::Frags;
    constexpr int TERMS = ide_terms(IDE_DEG);
    constexpr int kEnvFrags = EnvLayout<TERMS, (ENV_T ? ENV_T : 1)>::Frags;
    constexpr uint32_t kEnvChunks = pass_chunks(kEnvFrags), kHeadChunks = pass_chunks(kHeadFrags);
    constexpr int kEnvN = ring_padded(kEnvFrags), kHeadN = ring_padded(kHeadFrags);
    // ================= environment MLP on IDE(normal) and IDE(reflection) =====================
    float env_n[12];
    if constexpr (kEnvNet && PRE_ENV) {
        const float4* ep = reinterpret_cast<const float4*>(env_pre);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float4 u = ep[q], v = ep[3 + q];
            env_n[4 * q] = u.x; env_n[4 * q + 1] = u.y; env_n[4 * q + 2] = u.z; env_n[4 * q + 3] = u.w;
            env_r[4 * q] = v.x; env_r[4 * q + 1] = v.y; env_r[4 * q + 2] = v.z; env_r[4 * q + 3] = v.w;
        }
    }
    if constexpr (kEnvNet && !PRE_ENV)
#pragma unroll 1
    for (int enc = 0; enc < 2; ++enc) {
        const float vx = enc ? wr[0] : nenv[0], vy = enc ? wr[1] : nenv[1], vz = enc ? wr[2] : nenv[2];
        const float kinv = enc ? rough : c.kappa_diffuse;
        float code[2 * TERMS];
        ide_eval<IDE_DEG, true>(vx, vy, vz, kinv, [&](int j, float re, float im) {
            code[j] = re * c.light_scale;
            code[TERMS + j] = im * c.light_scale;
        });
#pragma unroll
        for (int s = 0; s < TERMS; ++s) pack_pair(code[2 * s], code[2 * s + 1]);
        tick(4);   // IDE
        f32x16 outA, outB;
#pragma unroll 1
        for (int grp = 0; grp < 2; ++grp) {
            float in[TERMS];
#pragma unroll
            for (int s = 0; s < TERMS; ++s) in[s] = grp ? code[2 * s + 1] : code[2 * s];
            f32x16 o;
            const bool last = enc == 1 && grp == 1;
            wp.begin_pass(c.env_blob, kEnvChunks, last ? c.head_blob : c.env_blob, last ? kHeadChunks : kEnvChunks);
            env_pass<TERMS, ENV_T, kEnvN, env_handoff(TERMS, ENV_T)>(wp, lane, aux, in, o);    // env_pass.hip.h
            if (grp == 0) outA = o; else outB = o;
        }
        tick(5);   // env mlp
        float e12[12];
        {
            float alo[4], ahi[4], blo[4], bhi[4];
            fold16(outA, alo, ahi);
            fold16(outB, blo, bhi);
            rows_to_lanes<3>(alo, ahi, blo, bhi, e12);
        }
        normalize_n<12>(e12, 1e-12f);                                               // network.py:541,600
        if (enc == 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) env_n[i] = e12[i];
        } else {
#pragma unroll
            for (int i = 0; i < 12; ++i) env_r[i] = e12[i];
        }
    }

    // ================= diffuse and specular heads ============================================
    {
        // env family : diffuse input [geo_feat | env(normal)] (24), specular input [geo_feat | normal | env(refl) | n.v] (28)
        // no-env family: diffuse input geo_feat (12), specular input [SH(d) | geo_feat | SH(normal) | n.v] (network.py:576-584)
        float din[2 * kDSteps], sin_[2 * kSSteps];
        if constexpr (kEnvNet) {
#pragma unroll
            for (int i = 0; i < 12; ++i) { din[i] = geo[i]; din[12 + i] = env_n[i]; sin_[i] = geo[i]; sin_[15 + i] = env_r[i]; }
            sin_[12] = nrm[0]; sin_[13] = nrm[1]; sin_[14] = nrm[2]; sin_[27] = ndot;
        } else {
            float shd[kShDim ? kShDim : 1], shn[kShDim ? kShDim : 1];
            sh_eval<(SH_DEG ? SH_DEG : 1), false>(vd[0], vd[1], vd[2], shd, nullptr, nullptr, nullptr);
            sh_eval<(SH_DEG ? SH_DEG : 1), false>(nrm[0], nrm[1], nrm[2], shn, nullptr, nullptr, nullptr);
#pragma unroll
            for (int i = 0; i < 12; ++i) { din[i] = geo[i]; sin_[kShDim + i] = geo[i]; }
#pragma unroll
            for (int i = 0; i < kShDim; ++i) { sin_[i] = shd[i]; sin_[kShDim + 12 + i] = shn[i]; }
            sin_[2 * kShDim + 12] = ndot;
#pragma unroll
            for (int i = kSpecIn; i < 2 * kSSteps; ++i) sin_[i] = 0;      // odd input width: zero pad (weights are zero there too)
        }
#pragma unroll
        for (int s = 0; s < kDSteps; ++s) pack_pair(din[2 * s], din[2 * s + 1]);
#pragma unroll
        for (int s = 0; s < kSSteps; ++s) pack_pair(sin_[2 * s], sin_[2 * s + 1]);
        f32x16 dA, dB, sA, sB;
#pragma unroll 1
        for (int grp = 0; grp < 2; ++grp) {
            float in_d[kDSteps], in_s[kSSteps];
#pragma unroll
            for (int s = 0; s < kDSteps; ++s) in_d[s] = grp ? din[2 * s + 1] : din[2 * s];
#pragma unroll
            for (int s = 0; s < kSSteps; ++s) in_s[s] = grp ? sin_[2 * s + 1] : sin_[2 * s];
            f32x16 d1[1], d2, s1[2], s2[2], s3;
            wp.begin_pass(c.head_blob, kHeadChunks, grp == 0 ? c.head_blob : c.next_blob, grp == 0 ? kHeadChunks : c.next_chunks);
            pipe_layer_from_lanes<kDSteps, 1, kHeadD1, kHeadN>(wp, lane, in_d, d1);
            pipe_layer16_from_tiles<1, kHeadD2, kHeadN, true>(wp, lane, d1, d2);
            pipe_layer_from_lanes<kSSteps, 2, kHeadS1, kHeadN>(wp, lane, in_s, s1);
            pipe_layer_from_tiles<2, 2, kHeadS2, kHeadN, true>(wp, lane, s1, s2);
            pipe_layer16_from_tiles<2, kHeadS3, kHeadN, true>(wp, lane, s2, s3);
            wp.template end_pass<kHeadFrags>();
            if (grp == 0) { dA = d2; sA = s3; } else { dB = d2; sB = s3; }
        }
        float alo[4], ahi[4], blo[4], bhi[4], rgb[4];
        fold16(dA, alo, ahi);
        fold16(dB, blo, bhi);
        rows_to_lanes<1>(alo, ahi, blo, bhi, rgb);
#pragma unroll
        for (int r = 0; r < 3; ++r) cd[r] = sigmoidf(rgb[r]);                       // color_act, metallic = 1
        fold16(sA, alo, ahi);
        fold16(sB, blo, bhi);
        rows_to_lanes<1>(alo, ahi, blo, bhi, rgb);
#pragma unroll
        for (int r = 0; r < 3; ++r) cs[r] = sigmoidf(rgb[r]);
    }

}

// ---- reflected radiance of the sample's ray -> 12 features -> specular head again -> learnt blend (network.py:612-659,683-690).
// `wp` must be streaming renv_blob on entry; on return it streams `after_blob`.  Shared by the persistent kernel and the
// record-shading kernel (third pass of indirect rendering).
template <class WP>
__device__ __forceinline__ void shade_renv(WP& wp, const uint32_t lane, const float* renv_blob, const float* spec2_blob, const float* after_blob,
                                           const uint32_t after_chunks, const float (&rimg)[4], const float rough, const float rough_scale,
                                           const float indir_rough_thresh, const float blend_logit, const float (&geo)[12],
                                           const float (&nrm)[3], const float ndot, float (&cs)[3]) {
    constexpr uint32_t kRenvChunks = pass_chunks(kRenvFrags), kSpec2Chunks = pass_chunks(kSpec2Frags);
    constexpr int kRenvN = ring_padded(kRenvFrags), kSpec2N = ring_padded(kSpec2Frags);
    const float vis = rimg[3];
    const float remap = sqrtf(rough / rough_scale / 0.75f);
    float rin[4] = {rimg[0] * vis, rimg[1] * vis, rimg[2] * vis, remap};
    pack_pair(rin[0], rin[1]);
    pack_pair(rin[2], rin[3]);
    f32x16 eA, eB;
#pragma unroll 1
    for (int grp = 0; grp < 2; ++grp) {
        float in[2] = {grp ? rin[1] : rin[0], grp ? rin[3] : rin[2]};
        f32x16 r1[2], r2[2], r3;
        wp.begin_pass(renv_blob, kRenvChunks, grp == 0 ? renv_blob : spec2_blob, grp == 0 ? kRenvChunks : kSpec2Chunks);
        pipe_layer_from_lanes<2, 2, kRenv1, kRenvN>(wp, lane, in, r1);
        pipe_layer_from_tiles<2, 2, kRenv2, kRenvN, true>(wp, lane, r1, r2);
        pipe_layer_from_tiles<2, 2, kRenv3, kRenvN, true>(wp, lane, r2, r1);
        pipe_layer16_from_tiles<2, kRenv4, kRenvN, true>(wp, lane, r1, r3);
        wp.template end_pass<kRenvFrags>();
        if (grp == 0) eA = r3; else eB = r3;
    }
    float e12[12];
    {
        float alo[4], ahi[4], blo[4], bhi[4];
        fold16(eA, alo, ahi);
        fold16(eB, blo, bhi);
        rows_to_lanes<3>(alo, ahi, blo, bhi, e12);
    }
    normalize_n<12>(e12, 1e-12f);
    float sin2[28];
#pragma unroll
    for (int i = 0; i < 12; ++i) { sin2[i] = geo[i]; sin2[15 + i] = e12[i]; }
    sin2[12] = nrm[0]; sin2[13] = nrm[1]; sin2[14] = nrm[2]; sin2[27] = ndot;
#pragma unroll
    for (int s = 0; s < 14; ++s) pack_pair(sin2[2 * s], sin2[2 * s + 1]);
    f32x16 cA, cB;
#pragma unroll 1
    for (int grp = 0; grp < 2; ++grp) {
        float in_s[14];
#pragma unroll
        for (int s = 0; s < 14; ++s) in_s[s] = grp ? sin2[2 * s + 1] : sin2[2 * s];
        f32x16 s1[2], s2[2], s3;
        wp.begin_pass(spec2_blob, kSpec2Chunks, grp == 0 ? spec2_blob : after_blob, grp == 0 ? kSpec2Chunks : after_chunks);
        pipe_layer_from_lanes<14, 2, kSpec2S1, kSpec2N>(wp, lane, in_s, s1);
        pipe_layer_from_tiles<2, 2, kSpec2S2, kSpec2N, true>(wp, lane, s1, s2);
        pipe_layer16_from_tiles<2, kSpec2S3, kSpec2N, true>(wp, lane, s2, s3);
        wp.template end_pass<kSpec2Frags>();
        if (grp == 0) cA = s3; else cB = s3;
    }
    const bool masked = rough < indir_rough_thresh && vis > 0.9f;
    const float blend = 0.98f * sigmoidf(blend_logit);                              // learn_indir_blend, network.py:443-446,630
    float alo[4], ahi[4], blo[4], bhi[4], rgb[4];
    fold16(cA, alo, ahi);
    fold16(cB, blo, bhi);
    rows_to_lanes<1>(alo, ahi, blo, bhi, rgb);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float c_renv = sigmoidf(rgb[r]);
        if (masked) cs[r] = cs[r] * blend + c_renv * (1 - blend);
    }
}

// Two network families:
//   SH_DEG == 0: environment-MLP family (toaster.ini / neural_renderer.ini): IDE degree IDE_DEG, env hidden 32 ENV_T
//   SH_DEG  > 0: no environment network (BASELINE configs[1]): diffuse head on geo_feat, specular head on
//                [SH(view dir) | geo_feat | SH(normal) | n.v] with SH "degree" SH_DEG (SH_DEG^2 values each)
// GEOM: geometry-only launches (first pass of indirect rendering, geometry pass of a two-phase frame) get their own
// instantiation without any shading code
template <int IDE_DEG, int ENV_T, int SH_DEG, bool GEOM = false>
__global__ void __launch_bounds__(kBlockThreads, 1) k_render_persistent(const RenderArgs a) {
    constexpr bool kEnvNet = SH_DEG == 0;
    constexpr int kShDim = SH_DEG * SH_DEG;
    constexpr int kDiffIn = kEnvNet ? 24 : 12, kSpecIn = kEnvNet ? 28 : 2 * kShDim + 13;
    constexpr int kDSteps = (kDiffIn + 1) / 2, kSSteps = (kSpecIn + 1) / 2;
    using Head = HeadLayout<kDSteps, kSSteps>;
    constexpr int kHeadFrags = Head::Frags;
    constexpr int TERMS = ide_terms(IDE_DEG);      // IDE_DIM = 2 * TERMS input features, TERMS lane-order steps
    constexpr int kEnv0 = 0, kEnv1 = kEnv0 + lane_layer_frags(TERMS, ENV_T, true), kEnv2 = kEnv1 + tile_layer_frags(ENV_T, ENV_T, true),
                  kEnv3 = kEnv2 + tile_layer_frags(ENV_T, ENV_T, true), kEnvFrags = kEnv3 + tile_layer_frags(ENV_T, 1, true);
    constexpr uint32_t kSdfChunks = pass_chunks(kSdfFrags), kEnvChunks = pass_chunks(kEnvFrags), kHeadChunks = pass_chunks(kHeadFrags);
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ float s_jac[kLevels * 6 * 64];
    // per-ray state that is only touched by march and blend, parked here while a sample is shaded (it would
    // otherwise sit in -- or be spilled from -- registers the 256-wide layers need)
    constexpr int kParked = 26;
    __shared__ float s_park[kParked * 64];
    // environment pass: ReLU staging slot + bias tiles (env_pass.hip.h)
    constexpr bool kEnvLds = SH_DEG == 0 && !GEOM;
    __shared__ __attribute__((aligned(16))) float s_env[kEnvLds ? EnvLayout<ide_terms(IDE_DEG), (ENV_T ? ENV_T : 1)>::kLdsFloats : 4];
    EnvAux aux = {nullptr, nullptr};
    if constexpr (kEnvLds) aux = env_lds_init<ide_terms(IDE_DEG), ENV_T>(a.env_blob, s_env, lane);
    WeightRing<kRingDepth> wp;
    wp.start(lane, a.sdf_blob, kSdfChunks);
    // fragments per pass as the weight source sees them (the ring pads every pass to a multiple of its depth)
    constexpr int kSdfN = ring_padded(kSdfFrags);

    // ---- per-lane ray slot ---------------------------------------------------------------------
    int ray = -1;
    bool drained = false;          // this lane saw the queue run dry
    uint32_t n_taken = 0;          // samples composited for the current ray
    RayGeom rg = {};
    float far = 0, t_ray = 0, t_resume = 0;
    const uint32_t n_hit = __builtin_amdgcn_readfirstlane(a.ray_counter[1]);
    Accum acc = {};
    float an[3] = {0, 0, 0}, ad[3] = {0, 0, 0}, as[3] = {0, 0, 0}, arough = 0;
    float rimg[4] = {0, 0, 0, 0};      // this ray's reflected radiance (rgb, visibility) when a.r_images is given
    unsigned long long n_samples = 0, n_rounds = 0, n_rays = 0;

    auto finish_ray = [&]() {
        const size_t id = (size_t)ray;
        const float rest = 1 - acc.ws;
        a.image[3 * id] = acc.r + rest * a.bg; a.image[3 * id + 1] = acc.g + rest * a.bg; a.image[3 * id + 2] = acc.b + rest * a.bg;
        a.depth[id] = acc.depth;
        a.ws[id] = acc.ws;
        if (a.normal) {
            float v[3] = {an[0], an[1], an[2]};
            const float inv = 1.0f / fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-10f);
            a.normal[3 * id] = v[0] * inv; a.normal[3 * id + 1] = v[1] * inv; a.normal[3 * id + 2] = v[2] * inv;
        }
        if (a.diffuse) { a.diffuse[3 * id] = ad[0]; a.diffuse[3 * id + 1] = ad[1]; a.diffuse[3 * id + 2] = ad[2]; }
        if (a.specular) { a.specular[3 * id] = as[0]; a.specular[3 * id + 1] = as[1]; a.specular[3 * id + 2] = as[2]; }
        if (a.roughness) a.roughness[id] = arough;
        if (a.ray_cost) a.ray_cost[id] = (uint16_t)min(n_taken, 65535u);
        ray = -1;
    };

    // ---- tail mode ---------------------------------------------------------------------------
    // Once the queue is empty a wave would finish its last rays one sample per round with most lanes
    // idle.  Instead the surviving rays are re-packed and each gets k = 2, 4, 8 adjacent lanes that
    // shade k CONSECUTIVE samples of it per round -- the reference's own n_step batching
    // (cuda_ray.py:287), applied per wave.  All k lanes keep an identical copy of the ray state.
    uint32_t k = 1;

    for (;;) {
        if (__any(drained)) {
            drained = true;                                  // the queue head only grows: empty for one is empty for all
            const uint32_t sub0 = lane & (k - 1);
            const unsigned long long leaders = __ballot(ray >= 0 && sub0 == 0);
            const uint32_t active = (uint32_t)__popcll(leaders);
            uint32_t nk = k;
            while (nk < kMaxGroup && active * nk * 2 <= 64) nk *= 2;
            if (nk != k && active > 0) {
                // lanes [r nk, (r+1) nk) adopt the r-th surviving ray, i.e. the state of the r-th set bit of `leaders`
                const uint32_t g = lane / nk;
                const bool on = g < active;
                const int src = on ? (int)nth_set_bit(leaders, g) : 0;
                const int ray_src = __shfl(ray, src);    // every lane must execute the shuffle (source lanes push their data)
                ray = on ? ray_src : -1;
                far = __shfl(far, src); t_ray = __shfl(t_ray, src); t_resume = __shfl(t_resume, src);
                n_taken = __shfl(n_taken, src);
                acc.ws = __shfl(acc.ws, src); acc.depth = __shfl(acc.depth, src); acc.t = __shfl(acc.t, src);
                acc.r = __shfl(acc.r, src); acc.g = __shfl(acc.g, src); acc.b = __shfl(acc.b, src);
#pragma unroll
                for (int d = 0; d < 3; ++d) { an[d] = __shfl(an[d], src); ad[d] = __shfl(ad[d], src); as[d] = __shfl(as[d], src); }
                arough = __shfl(arough, src);
                if (ray >= 0) {
                    rg = load_ray(a.rays_o, a.rays_d, (uint32_t)ray);
                    if (a.r_images) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) rimg[j] = a.r_images[4 * (size_t)ray + j];
                    }
                }
                k = nk;
            }
        }
        const uint32_t sub = lane & (k - 1);

        // ================= refill + march: every lane ends with a sample or idle =================
        bool have = false;
        float px = 0, py = 0, pz = 0, dt = 0, delta_depth = 0, t_next = 0;
        if (k > 1) {
            // lane `sub` of a group takes the (sub+1)-th next occupied sample of the group's ray
            if (ray >= 0) {
                float t = n_taken == 0 ? t_resume : t_ray;
                float last_t = t_ray;
                bool ok = true;
                for (uint32_t i = 0; i <= sub && ok; ++i) {
                    if (i) last_t = t;
                    ok = (n_taken + i) < a.max_samples && march_next(a.mk, rg, far, t, px, py, pz, dt);
                }
                have = ok;
                delta_depth = t - last_t;
            }
            // a ray whose NEXT sample does not exist is finished here, so that every ray still open
            // after this point has at least its leader's sample
            const bool lead_have = __shfl((int)have, (int)(lane & ~(k - 1))) != 0;
            if (ray >= 0 && !lead_have) {
                if (sub == 0) finish_ray(); else ray = -1;
                have = false;
            }
        } else
        for (;;) {
            if (ray < 0 && !drained) {
                const uint32_t slot = atomicAdd(a.ray_counter, 1u);     // aggregated to one atomic per wave
                if (slot < n_hit) {
                    const uint32_t id = a.hit_ids[slot];
                    ray = (int)id;
                    rg = load_ray(a.rays_o, a.rays_d, id);
                    if (a.r_images) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) rimg[j] = a.r_images[4 * (size_t)id + j];
                    }
                    float near;
                    near_far(rg, a.box, a.min_near, near, far);
                    t_ray = near;
                    t_resume = a.hit_t[id];
                    n_taken = 0;
                    acc.ws = acc.depth = acc.r = acc.g = acc.b = 0; acc.t = near;
                    an[0] = an[1] = an[2] = ad[0] = ad[1] = ad[2] = as[0] = as[1] = as[2] = arough = 0;
                    ++n_rays;
                } else {
                    drained = true;
                }
            }
            if (ray >= 0 && !have) {
                // a freshly fetched ray jumps to the first sample found by the pre-pass; the depth
                // delta is still measured from where the reference marcher would have started (near)
                float t = n_taken == 0 ? t_resume : t_ray;
                const float last_t = t_ray;
                if (n_taken < a.max_samples && march_next(a.mk, rg, far, t, px, py, pz, dt)) {
                    have = true;
                    delta_depth = t - last_t;
                    t_next = t;
                } else {
                    finish_ray();
                }
            }
            const bool again = !have && ray < 0 && !drained;
            if (!__any(again)) break;
        }
        // the four waves share the weight pipe: the block keeps going while any of them has a sample
        if (!__any(have)) break;
   // refill + march
        n_samples += have ? 1 : 0;
        n_rounds += 1;
        if (!have) { px = py = pz = 0; }
        float* park = s_park + lane;
        {
            const float v[kParked] = {rg.ox, rg.oy, rg.oz, rg.rdx, rg.rdy, rg.rdz, far, t_ray, t_resume, acc.ws, acc.depth, acc.r, acc.g,
                                      acc.b, acc.t, an[0], an[1], an[2], ad[0], ad[1], ad[2], as[0], as[1], as[2], arough,
                                      __uint_as_float(n_taken)};
#pragma unroll
            for (int i = 0; i < kParked; ++i) park[i * 64] = v[i];
            asm volatile("" ::: "memory");      // the values must really leave the registers: no store-to-load forwarding
        }

        // ================= hash grid: features + Jacobian (per lane) ============================
        // Software-pipelined over levels: the 8 corner gathers of level l+2 are issued before level l is
        // interpolated, so two levels' worth of (Infinity-Cache) gather latency hides under the VALU work
        // of the current one.  Arithmetic per level is eval_level's (grid_core.hip.h), op for op.
        // The 96 Jacobian entries per sample are parked in LDS ([entry][lane]: conflict-free) until the SDF
        // backward pass needs them; keeping them in VGPRs across the SDF network spills.
        float feat[2 * kLevels];
        float* jac_col = s_jac + lane;
        {
            // (xyz + bound) / (2 bound)  -- hashencoder/hashgrid.py:161
            const float x01[3] = {(px + a.mk.bound) / a.bound2, (py + a.mk.bound) / a.bound2, (pz + a.mk.bound) / a.bound2};
            const bool inside = x01[0] >= 0 && x01[0] <= 1 && x01[1] >= 0 && x01[1] <= 1 && x01[2] >= 0 && x01[2] <= 1;
            // outside the unit cube every level contributes zeros (hashencoder.cu:124-149); evaluating
            // at a clamped position and masking afterwards keeps the gathers in bounds
            const float xc[3] = {inside ? x01[0] : 0.5f, inside ? x01[1] : 0.5f, inside ? x01[2] : 0.5f};
            HashStage st[kHashAhead + 1];      // rotating stages; every index below is a compile-time constant
            [&]<int... I>(std::integer_sequence<int, I...>) {
                (hash_prepare(a.lv[I], a.table, xc, st[I]), ...);
            }(std::make_integer_sequence<int, kHashAhead>{});
            __builtin_amdgcn_sched_barrier(0);
            auto level = [&](auto lc, HashStage& cur, HashStage& ahead) {
                constexpr int l = decltype(lc)::value;
                if constexpr (l + kHashAhead < kLevels) hash_prepare(a.lv[l + kHashAhead], a.table, xc, ahead);
                __builtin_amdgcn_sched_barrier(0);
                float o[2], g[3][2];
                hash_finish(a.lv[l].scale, cur, o, g);
                const float m = (a.lv[l].enabled && inside) ? 1.0f : 0.0f;     // network.py:390-393 level mask
                feat[2 * l] = o[0] * m; feat[2 * l + 1] = o[1] * m;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    jac_col[((l * 3 + d) * 2 + 0) * 64] = g[d][0] * m;
                    jac_col[((l * 3 + d) * 2 + 1) * 64] = g[d][1] * m;
                }
            };
            [&]<int... L>(std::integer_sequence<int, L...>) {
                (level(std::integral_constant<int, L>{}, st[L % (kHashAhead + 1)], st[(L + kHashAhead) % (kHashAhead + 1)]), ...);
            }(std::make_integer_sequence<int, kLevels>{});
        }

   // hash grid
        // ================= SDF network forward + input gradient (matrix cores) ===================
        float h3[32];      // raw outputs of the last SDF layer for this lane's sample (rows 0..14 used)
        float gfeat[32];   // d sdf / d feat
        {
            float inA[kLevels], inB[kLevels];
#pragma unroll
            for (int s = 0; s < kLevels; ++s) {
                float e = feat[2 * s], o = feat[2 * s + 1];
                pack_pair(e, o);
                inA[s] = e; inB[s] = o;
            }
            f32x16 outA, outB, gfA, gfB;
#pragma unroll 1
            for (int grp = 0; grp < 2; ++grp) {
                float in[kLevels];
#pragma unroll
                for (int s = 0; s < kLevels; ++s) in[s] = grp ? inB[s] : inA[s];
                f32x16 h1[2], h2[2], o3;
                const bool to_sdf = grp == 0 || GEOM || a.geometry_only;       // geometry-only: the SDF blob is the only one streamed
                wp.begin_pass(a.sdf_blob, kSdfChunks, to_sdf ? a.sdf_blob : (kEnvNet ? a.env_blob : a.head_blob),
                              to_sdf ? kSdfChunks : (kEnvNet ? kEnvChunks : kHeadChunks));
                // forward; of the two hidden pre-activations only their signs are needed again (the ReLU masks of the backward
                // pass): they are kept as one bit per feature (bit 16 t + r of a lane = register r of tile t), which frees 64
                // registers while the backward layers run
                uint32_t pos1 = 0, pos2 = 0;
                pipe_layer_from_lanes<kLevels, 2, kSdfW1, kSdfN>(wp, lane, in, h1);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) pos1 |= (h1[t][r] > 0 ? 1u : 0u) << (16 * t + r);
                pipe_layer_from_tiles<2, 2, kSdfW2, kSdfN, true>(wp, lane, h1, h2);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) pos2 |= (h2[t][r] > 0 ? 1u : 0u) << (16 * t + r);
                pipe_layer16_from_tiles<2, kSdfW3, kSdfN, true>(wp, lane, h2, o3);       // 64 -> 15 on 16-row MFMA blocks (k_order 2)
                // backward of sdf = o3[row 0]:  g2 = W3[0,:] * [h2 > 0];  g1 = (W2^T g2) * [h1 > 0];  gfeat = W1^T g1
                f32x16 g2[2], g1[2], gf[1];
                const ParamBuf w3r0 = make_param_buf(a.sdf_w3r0, 2 * 128u, lane);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x16 w = load_rowvec(w3r0, t);
#pragma unroll
                    for (int r = 0; r < 16; ++r) g2[t][r] = (pos2 >> (16 * t + r)) & 1u ? w[r] : 0.0f;
                }
                pipe_layer_from_tiles<2, 2, kSdfW2t, kSdfN, false, false>(wp, lane, g2, g1);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) g1[t][r] = (pos1 >> (16 * t + r)) & 1u ? g1[t][r] : 0.0f;
                pipe_layer_from_tiles<2, 1, kSdfW1t, kSdfN, false, false>(wp, lane, g1, gf);
                wp.template end_pass<kSdfFrags>();
                if (grp == 0) { outA = o3; gfA = gf[0]; } else { outB = o3; gfB = gf[0]; }
            }
            {
                float alo[4], ahi[4], blo[4], bhi[4];
                fold16(outA, alo, ahi);
                fold16(outB, blo, bhi);
                float h16[16];
                rows_to_lanes<4>(alo, ahi, blo, bhi, h16);
#pragma unroll
                for (int r = 0; r < 16; ++r) h3[r] = h16[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = gfA[r], q = gfB[r];
                unpack_pair(p, q);
                gfeat[tile_row(r, 0)] = p; gfeat[tile_row(r, 1)] = q;
            }
        }

   // sdf mlp fwd+bwd
        // ================= per-sample geometry terms =============================================
        const float sdf = h3[0];
        float geo[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) geo[i] = h3[1 + i];
        normalize_n<12>(geo, 1e-12f);                                                   // network.py:434-435
        const float rough = a.rough_act_scale * softplusf(h3[13] + a.rough_bias) * a.rough_scale;   // network.py:443-448
        // kernel_input_backward order: levels outer, channels inner (hashencoder.cu:346-372)
        float nrm[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float s = 0;
#pragma unroll
            for (int l = 0; l < kLevels; ++l) {
                s += gfeat[2 * l] * jac_col[((l * 3 + d) * 2 + 0) * 64];
                s += gfeat[2 * l + 1] * jac_col[((l * 3 + d) * 2 + 1) * 64];
            }
            nrm[d] = s / a.bound2;                                                      // d x01 / d xyz
        }
        normalize_n<3>(nrm, 1e-10f);                                                    // renderer.py:192
        // Laplace density (network.py:32-37): (1/beta) (0.5 + 0.5 sign(s) expm1(-|s| / beta))
        const float sgn = sdf > 0 ? 1.0f : (sdf < 0 ? -1.0f : 0.0f);
        const float sigma = a.inv_beta * (0.5f + 0.5f * sgn * expm1f(-fabsf(sdf) / a.beta)) * a.density_scale;

        // view-dependent inputs (renderer.py:147-180)
        const float wo[3] = {-rg.dx, -rg.dy, -rg.dz};
        const float ndot = nrm[0] * wo[0] + nrm[1] * wo[1] + nrm[2] * wo[2];
        float wr[3], nenv[3] = {nrm[0], nrm[1], nrm[2]};
        {
            const float c = 2 * ndot;                                                   // reflect_dir, renderer.py:38
#pragma unroll
            for (int d = 0; d < 3; ++d) wr[d] = c * nrm[d] - wo[d];
            if (a.has_rot) {
                const float w0 = wr[0], w1 = wr[1], w2 = wr[2], n0 = nenv[0], n1 = nenv[1], n2 = nenv[2];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    wr[j] = w0 * a.rot[j] + w1 * a.rot[3 + j] + w2 * a.rot[6 + j];
                    nenv[j] = n0 * a.rot[j] + n1 * a.rot[3 + j] + n2 * a.rot[6 + j];
                }
            }
        }

   // geometry terms
        // ================= shading: environment MLP x2 + diffuse / specular heads ===================
        float cd[3] = {0, 0, 0}, cs[3] = {0, 0, 0};
        if constexpr (!GEOM) if (!a.geometry_only) {
            const bool renv = kEnvNet && a.r_images != nullptr;
            constexpr uint32_t kRenvChunks = pass_chunks(kRenvFrags);
            const ShadeConsts sc = {a.env_blob, a.head_blob, renv ? a.renv_blob : a.sdf_blob, renv ? kRenvChunks : kSdfChunks,
                                    a.kappa_diffuse, a.light_scale};
            const float vd[3] = {rg.dx, rg.dy, rg.dz};
            float env_r[12];
            shade_sample<IDE_DEG, ENV_T, SH_DEG>(wp, lane, sc, nrm, nenv, wr, vd, ndot, geo, rough, cd, cs, env_r, [](int) {}, aux);
            if constexpr (kEnvNet) if (renv)
                shade_renv(wp, lane, a.renv_blob, a.spec2_blob, a.sdf_blob, kSdfChunks, rimg, rough, a.rough_scale, a.indir_rough_thresh,
                           h3[14], geo, nrm, ndot, cs);
        }
   // heads (+ env unpack)
        {
            asm volatile("" ::: "memory");
            float v[kParked];
#pragma unroll
            for (int i = 0; i < kParked; ++i) v[i] = park[i * 64];
            rg.ox = v[0]; rg.oy = v[1]; rg.oz = v[2]; rg.rdx = v[3]; rg.rdy = v[4]; rg.rdz = v[5]; far = v[6]; t_ray = v[7];
            t_resume = v[8]; acc.ws = v[9]; acc.depth = v[10]; acc.r = v[11]; acc.g = v[12]; acc.b = v[13]; acc.t = v[14];
            an[0] = v[15]; an[1] = v[16]; an[2] = v[17]; ad[0] = v[18]; ad[1] = v[19]; ad[2] = v[20]; as[0] = v[21]; as[1] = v[22];
            as[2] = v[23]; arough = v[24]; n_taken = __float_as_uint(v[25]);
        }
        // ================= composite (raymarching.cu:996-1030 recurrence) =========================
        bool ex_on = false;          // this lane's sample was composited (-> one geometry record when exporting)
        float ex_w = 0;
        uint32_t ex_idx = 0, ex_ray = 0;
        if (k == 1) {
            if (have) {
                const float alpha = 1.0f - expf(-sigma * dt);
                const float T = 1 - acc.ws;
                const float w = alpha * T;
                acc.ws += w;
                acc.t = acc.t + delta_depth;
                acc.depth += w * acc.t;
                const float cr = (cd[0] + cs[0]) * a.intensity_scale, cg = (cd[1] + cs[1]) * a.intensity_scale,
                            cb = (cd[2] + cs[2]) * a.intensity_scale;
                acc.r += w * cr; acc.g += w * cg; acc.b += w * cb;
#pragma unroll
                for (int d = 0; d < 3; ++d) { an[d] += w * nrm[d]; ad[d] += w * cd[d]; as[d] += w * cs[d]; }
                arough += w * rough;
                t_ray = acc.t;            // the reference resumes the marcher from the composited time
                (void)t_next;
                ex_on = true; ex_w = w; ex_idx = n_taken; ex_ray = (uint32_t)ray;
                ++n_taken;
                if (T < a.T_thresh) finish_ray();
            }
        } else {
            // the k samples of a group are composited in order by EVERY lane of the group (identical
            // copies of the ray state); sample j comes from lane leader + j
            const int leader = (int)(lane & ~(k - 1));
            const float alpha_own = 1.0f - expf(-sigma * dt);
            bool open = ray >= 0;
            for (uint32_t j = 0; j < k; ++j) {
                const int src = leader + (int)j;
                const bool have_j = __shfl((int)have, src) != 0;
                const float alpha = __shfl(alpha_own, src), dd = __shfl(delta_depth, src), rg_j = __shfl(rough, src);
                float cdj[3], csj[3], nj[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) { cdj[d] = __shfl(cd[d], src); csj[d] = __shfl(cs[d], src); nj[d] = __shfl(nrm[d], src); }
                if (open) {
                    if (!have_j) {                       // the ray left the scene before its j-th sample
                        if (sub == 0) finish_ray(); else ray = -1;
                        open = false;
                    } else {
                        const float T = 1 - acc.ws;
                        const float w = alpha * T;
                        acc.ws += w;
                        acc.t = acc.t + dd;
                        acc.depth += w * acc.t;
                        acc.r += w * ((cdj[0] + csj[0]) * a.intensity_scale);
                        acc.g += w * ((cdj[1] + csj[1]) * a.intensity_scale);
                        acc.b += w * ((cdj[2] + csj[2]) * a.intensity_scale);
#pragma unroll
                        for (int d = 0; d < 3; ++d) { an[d] += w * nj[d]; ad[d] += w * cdj[d]; as[d] += w * csj[d]; }
                        arough += w * rg_j;
                        if (sub == j) { ex_on = true; ex_w = w; ex_idx = n_taken; ex_ray = (uint32_t)ray; }   // this lane shaded sample j
                        ++n_taken;
                        if (T < a.T_thresh) {
                            if (sub == 0) finish_ray(); else ray = -1;
                            open = false;
                        }
                    }
                }
            }
            if (open) t_ray = acc.t;
        }
        if (a.ex_counter) {
            // append one record per composited sample: wave-aggregated slot allocation, any order (the host sorts by (ray, idx))
            const unsigned long long em = __ballot(ex_on);
            uint32_t base = 0;
            if (lane == 0 && em) base = atomicAdd(a.ex_counter, (uint32_t)__popcll(em));
            base = __shfl(base, 0);
            const uint32_t slot = base + (uint32_t)__popcll(em & ((1ull << lane) - 1ull));
            if (ex_on && slot < a.ex_capacity) {
                a.ex_ray[slot] = ex_ray; a.ex_idx[slot] = ex_idx; a.ex_w[slot] = ex_w; a.ex_rough[slot] = rough;
#pragma unroll
                for (int d = 0; d < 3; ++d) a.ex_normal[3 * (size_t)slot + d] = nrm[d];
#pragma unroll
                for (int i = 0; i < 12; ++i) a.ex_geo[12 * (size_t)slot + i] = geo[i];
            }
        }
   // composite
    }

    if (a.stats) {
        // wave-level reduction then one atomic per counter
        for (int off = 32; off > 0; off >>= 1) {
            n_samples += __shfl_down(n_samples, off);
            n_rays += __shfl_down(n_rays, off);
        }
        if (lane == 0) {
            if (blockIdx.x == 0 && wave == 0) atomicAdd(&a.stats[2], (unsigned long long)(a.N - n_hit));   // rays finished by the pre-pass
            atomicAdd(&a.stats[0], n_samples);
            atomicAdd(&a.stats[1], n_rounds);
            atomicAdd(&a.stats[2], n_rays);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Shade-only kernel: the shading half of the loop for samples whose geometry is already known
// (surface rendering as in demo.ipynb cell 17, re-lighting / env rotation of cached geometry).
// ------------------------------------------------------------------------------------------

// SH_DEG > 0: the no-environment family (heads only; SH-encoded view direction and normal, BASELINE configs[1]).
// RENV: the reflected-radiance branch on top of the environment family (third pass of indirect rendering; record mode only).
template <int IDE_DEG, int ENV_T, int SH_DEG = 0, bool RENV = false, bool PRE_ENV = false>
__global__ void __launch_bounds__(kBlockThreads, 1) k_shade_samples(const ShadeArgs a) {
    constexpr bool kEnvNet = SH_DEG == 0;
    constexpr int kShDim = SH_DEG * SH_DEG;
    constexpr int kDSteps = ((kEnvNet ? 24 : 12) + 1) / 2, kSSteps = ((kEnvNet ? 28 : 2 * kShDim + 13) + 1) / 2;
    constexpr int TERMS = ide_terms(IDE_DEG);
    constexpr int kEnvFrags = lane_layer_frags(TERMS, ENV_T, true) + 2 * tile_layer_frags(ENV_T, ENV_T, true) + tile_layer_frags(ENV_T, 1, true);
    constexpr uint32_t kEnvChunks = pass_chunks(kEnvFrags), kHeadChunks = pass_chunks(HeadLayout<kDSteps, kSSteps>::Frags);
    constexpr uint32_t kRenvChunks = pass_chunks(kRenvFrags);
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr bool kEnvLds = kEnvNet && !PRE_ENV;
    __shared__ __attribute__((aligned(16))) float s_env[kEnvLds ? EnvLayout<TERMS, (ENV_T ? ENV_T : 1)>::kLdsFloats : 4];
    EnvAux aux = {nullptr, nullptr};
    if constexpr (kEnvLds) aux = env_lds_init<TERMS, ENV_T>(a.env_blob, s_env, lane);
    WeightRing<kRingDepth> wp;
    // the first blob a round streams: the environment MLP, or the heads when there is none
    const float* first_blob = (kEnvNet && !PRE_ENV) ? a.env_blob : a.head_blob;
    constexpr uint32_t kFirstChunks = (kEnvNet && !PRE_ENV) ? kEnvChunks : kHeadChunks;
    wp.start(lane, first_blob, kFirstChunks);
    const ShadeConsts sc = {a.env_blob, a.head_blob, RENV ? a.renv_blob : first_blob, RENV ? kRenvChunks : kFirstChunks, a.kappa_diffuse, a.light_scale};
    const uint32_t waves = gridDim.x * (kBlockThreads / 64);
    // a frame whose records did not fit (count beyond the capacity) is not shaded at all: the host redoes it
    uint32_t M = a.M;
    if (a.m_dev) { const uint32_t md = __builtin_amdgcn_readfirstlane(*a.m_dev); M = md > a.M ? 0u : md; }
    const uint32_t* list = (a.list && (uint32_t)__builtin_amdgcn_readfirstlane(*a.m_all) != M) ? a.list : nullptr;
    // Rounds are CLAIMED, not assigned: a wave takes the next 64 records off a device counter when it starts a round (the claim for
    // the round after is issued at once, so its latency hides behind 280 us of work).  With a static grid stride every wave does
    // 117 or 118 rounds of the headline frame and the kernel lasts as long as its slowest wave -- the waves of the odd XCDs are
    // 1.35 % slower than those of the even ones on this chip (tools/probe/wave_times.py: mean wave 32.99 ms, kernel 33.42 ms);
    // claimed rounds end within one round of each other.  Which wave shades a record changes nothing about its result.
    // (claiming the tail in half rounds -- one 32-sample group -- was measured: the run-time group count costs the whole kernel more
    //  than the half round it saves at the end, 33.46 against 33.25 ms)
    auto claim = [&]() -> uint32_t {
        uint32_t b = 0;
        if (lane == 0) b = atomicAdd(a.work, 64u);
        return __builtin_amdgcn_readfirstlane(b);
    };
    uint32_t base = a.work ? claim() : (blockIdx.x * (kBlockThreads / 64) + wave) * 64;
    uint32_t ahead = a.work ? claim() : base + waves * 64;
    for (; base < M; base = ahead, ahead = a.work ? claim() : ahead + waves * 64) {
        const uint32_t id = base + lane;
        const bool on = id < M;
        const size_t i = on ? (list ? (size_t)list[id] : (size_t)id) : 0;
        const size_t gi = a.slot ? (size_t)a.slot[i] : i;          // where this record's geometry lives
        float nrm[3], vd[3], geo[12];
        const size_t ray = a.ray_ids ? (size_t)a.ray_ids[i] : 0;
        const float* dir = a.ray_ids ? a.rays_d + 3 * ray : a.dirs + 3 * i;
#pragma unroll
        for (int d = 0; d < 3; ++d) { nrm[d] = on ? a.normals[3 * gi + d] : 0.0f; vd[d] = on ? dir[d] : 0.0f; }
#pragma unroll
        for (int j = 0; j < 12; ++j) geo[j] = a.geo_feat[(size_t)a.geo_stride * gi + j];
        const float rough = a.roughness[(size_t)a.rough_stride * gi];
        // renderer.py:147-180 (same statements as the persistent kernel)
        const float wo[3] = {-vd[0], -vd[1], -vd[2]};
        const float ndot = nrm[0] * wo[0] + nrm[1] * wo[1] + nrm[2] * wo[2];
        float wr[3], nenv[3] = {nrm[0], nrm[1], nrm[2]};
        const float c2 = 2 * ndot;
#pragma unroll
        for (int d = 0; d < 3; ++d) wr[d] = c2 * nrm[d] - wo[d];
        if (a.has_rot) {
            const float w0 = wr[0], w1 = wr[1], w2 = wr[2], n0 = nenv[0], n1 = nenv[1], n2 = nenv[2];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                wr[j] = w0 * a.rot[j] + w1 * a.rot[3 + j] + w2 * a.rot[6 + j];
                nenv[j] = n0 * a.rot[j] + n1 * a.rot[3 + j] + n2 * a.rot[6 + j];
            }
        }
        float cd[3], cs[3], env_r[12];
        shade_sample<IDE_DEG, ENV_T, SH_DEG, PRE_ENV>(wp, lane, sc, nrm, nenv, wr, vd, ndot, geo, rough, cd, cs, env_r, [](int) {}, aux,
                                                      PRE_ENV ? a.env_pre + 24 * i : nullptr);
        if constexpr (RENV) {
            float rimg[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) rimg[j] = on ? a.r_images[4 * ray + j] : 0.0f;
            shade_renv(wp, lane, a.renv_blob, a.spec2_blob, first_blob, kFirstChunks, rimg, rough, a.rough_scale, a.indir_rough_thresh,
                       a.blend[gi], geo, nrm, ndot, cs);
        }
        if (on) {
#pragma unroll
            for (int d = 0; d < 3; ++d) { a.c_diffuse[3 * i + d] = cd[d]; a.c_specular[3 * i + d] = cs[d]; }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The environment MLP as a standalone operator: y = env_net(x) for rows of IDE codes (network.py:533-536, 592-595 -- the
// `for l in range(num_layers_env)` Linear / ReLU chains), 64 rows per wave round, exactly the pass the shading kernels run.
// For callers that keep the reference's operator loop (encoders and MLPs as separate calls): its torch form is four GEMMs and
// three ReLU passes over [M, 256] activations.
// ------------------------------------------------------------------------------------------
template <int IDE_DEG, int ENV_T>
__global__ void __launch_bounds__(kBlockThreads, 1) k_env_mlp(const float* __restrict__ env_blob, const float* __restrict__ x, float* __restrict__ y,
                                                               uint32_t M, uint32_t* __restrict__ work, uint32_t aligned) {
    constexpr int TERMS = ide_terms(IDE_DEG), IN = 2 * TERMS;
    using L = EnvLayout<TERMS, ENV_T>;
    constexpr uint32_t kEnvChunks = pass_chunks(L::Frags);
    constexpr int kEnvN = ring_padded(L::Frags);
    const uint32_t lane = lane_id();
    __shared__ __attribute__((aligned(16))) float s_env[L::kLdsFloats];
    __shared__ __attribute__((aligned(16))) float s_io[wave_tile_floats<IN>()];
    const EnvAux aux = env_lds_init<TERMS, ENV_T>(env_blob, s_env, lane);
    WeightRing<kRingDepth> wp;
    wp.start(lane, env_blob, kEnvChunks);
    auto claim = [&]() -> uint32_t {
        uint32_t b = 0;
        if (lane == 0) b = atomicAdd(work, 64u);
        return __builtin_amdgcn_readfirstlane(b);
    };
    uint32_t base = claim(), ahead = claim();
    for (; base < M; base = ahead, ahead = claim()) {
        const uint32_t rows = min(64u, M - base);
        float code[IN];
        if (aligned) wave_load_rows<IN, true>(s_io, code, x + (size_t)base * IN, rows, lane);
        else wave_load_rows<IN, false>(s_io, code, x + (size_t)base * IN, rows, lane);
        // the second group's operands wait in the (now free) row tile instead of in 36 registers beside the first group's pass
        float in[TERMS];
#pragma unroll
        for (int s = 0; s < TERMS; ++s) {
            pack_pair(code[2 * s], code[2 * s + 1]);
            s_io[s * 64 + lane] = code[2 * s + 1];
            in[s] = code[2 * s];
        }
        f32x16 outA, outB;
#pragma unroll 1
        for (int grp = 0; grp < 2; ++grp) {
            if (grp) {
#pragma unroll
                for (int s = 0; s < TERMS; ++s) in[s] = s_io[s * 64 + lane];
            }
            f32x16 o;
            wp.begin_pass(env_blob, kEnvChunks, env_blob, kEnvChunks);
            env_pass<TERMS, ENV_T, kEnvN, env_handoff(TERMS, ENV_T)>(wp, lane, aux, in, o);
            if (grp == 0) outA = o; else outB = o;
        }
        __builtin_amdgcn_wave_barrier();
        float e12[12];
        {
            float alo[4], ahi[4], blo[4], bhi[4];
            fold16(outA, alo, ahi);
            fold16(outB, blo, bhi);
            rows_to_lanes<3>(alo, ahi, blo, bhi, e12);
        }
        if (aligned) wave_store_rows<12, true>(s_io, e12, y + (size_t)base * 12, rows, lane);
        else wave_store_rows<12, false>(s_io, e12, y + (size_t)base * 12, rows, lane);
    }
}

// Composite shaded colours of cached geometry: one lane per ray, its samples contiguous and in march order, the
// compositing weights already known (they depend on geometry only).  Same accumulation order and arithmetic as the
// blend section of the persistent kernel.
__global__ void __launch_bounds__(kBlock) k_composite_shaded(const uint32_t* __restrict__ offsets, const float* __restrict__ w,
                                                             const float* __restrict__ cd, const float* __restrict__ cs,
                                                             const float* __restrict__ ws, uint32_t N, float intensity, float bg,
                                                             float* __restrict__ image, float* __restrict__ diffuse,
                                                             float* __restrict__ specular) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    float ar = 0, ag = 0, ab = 0, d0 = 0, d1 = 0, d2 = 0, s0 = 0, s1 = 0, s2 = 0;
    for (uint32_t i = offsets[r]; i < offsets[r + 1]; ++i) {
        const float wi = w[i];
        const float c0 = cd[3 * (size_t)i], c1 = cd[3 * (size_t)i + 1], c2 = cd[3 * (size_t)i + 2];
        const float e0 = cs[3 * (size_t)i], e1 = cs[3 * (size_t)i + 1], e2 = cs[3 * (size_t)i + 2];
        ar += wi * ((c0 + e0) * intensity); ag += wi * ((c1 + e1) * intensity); ab += wi * ((c2 + e2) * intensity);
        d0 += wi * c0; d1 += wi * c1; d2 += wi * c2;
        s0 += wi * e0; s1 += wi * e1; s2 += wi * e2;
    }
    const float rest = 1 - ws[r];
    image[3 * (size_t)r] = ar + rest * bg; image[3 * (size_t)r + 1] = ag + rest * bg; image[3 * (size_t)r + 2] = ab + rest * bg;
    if (diffuse) { diffuse[3 * (size_t)r] = d0; diffuse[3 * (size_t)r + 1] = d1; diffuse[3 * (size_t)r + 2] = d2; }
    if (specular) { specular[3 * (size_t)r] = s0; specular[3 * (size_t)r + 1] = s1; specular[3 * (size_t)r + 2] = s2; }
}

// Two-phase frames: records sit in the order the geometry pass appended them; ray r's samples are found through
// perm[offsets[r] + idx] = record (offsets = exclusive prefix sum of the per-ray sample counts).
// The records worth shading: compositing weight not exactly zero, IN RECORD ORDER (list[0] = their number, list[1..] = their
// indices; the block counts of the scan live behind the list).  An unordered gather (one atomic per wave) was measured 4 % slower
// on the whole shading pass although nothing was skipped: neighbouring shading waves then work on far-apart 64-record chunks of
// the ~1 GB of sample arrays instead of walking them front to back.
constexpr uint32_t kGatherChunk = 1024;       // records per workgroup of the gather kernels
__global__ void __launch_bounds__(kBlock) k_count_weighted_records(const float* __restrict__ w, const uint32_t* __restrict__ m_dev, uint32_t capacity,
                                                                   uint32_t* __restrict__ counts) {
    __shared__ uint32_t s_n;
    const uint32_t M = *m_dev > capacity ? 0u : *m_dev;        // a frame that did not fit: nothing is shaded, the host redoes it
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    uint32_t n = 0;
    for (uint32_t k = 0; k < kGatherChunk / kBlock; ++k) {
        const uint32_t i = blockIdx.x * kGatherChunk + k * kBlock + threadIdx.x;
        n += (i < M && w[i] != 0.0f) ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off);
    if ((threadIdx.x & 63u) == 0 && n) atomicAdd(&s_n, n);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_n;
}
// exclusive prefix of the block counts in place (one workgroup), total -> list[0]
__global__ void __launch_bounds__(1024) k_scan_record_counts(uint32_t* __restrict__ counts, uint32_t n_blocks, uint32_t* __restrict__ total) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < n_blocks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_blocks ? counts[i] : 0u;
        uint32_t incl = v;
        for (int off = 1; off < 64; off <<= 1) { const uint32_t up = __shfl_up(incl, off); if ((int)lane >= off) incl += up; }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = s_carry;
        for (uint32_t k = 0; k < wave; ++k) before += s_wave[k];
        if (i < n_blocks) counts[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}
__global__ void __launch_bounds__(kBlock) k_scatter_weighted_records(const float* __restrict__ w, const uint32_t* __restrict__ m_dev, uint32_t capacity,
                                                                     const uint32_t* __restrict__ counts, uint32_t* __restrict__ list) {
    __shared__ uint32_t s_wave[kBlock / 64];
    const uint32_t M = *m_dev > capacity ? 0u : *m_dev;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t at = counts[blockIdx.x];
    for (uint32_t k = 0; k < kGatherChunk / kBlock; ++k) {
        const uint32_t i = blockIdx.x * kGatherChunk + k * kBlock + threadIdx.x;
        const bool keep = i < M && w[i] != 0.0f;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = at;
        for (uint32_t q = 0; q < wave; ++q) before += s_wave[q];
        if (keep) list[1 + before + __popcll(m & ((1ull << lane) - 1ull))] = i;
        for (uint32_t q = 0; q < kBlock / 64; ++q) at += s_wave[q];
        __syncthreads();
    }
}

__global__ void __launch_bounds__(kBlock) k_place_records(const uint32_t* __restrict__ ray, const uint32_t* __restrict__ idx,
                                                          const uint32_t* __restrict__ m_dev, uint32_t capacity,
                                                          const uint32_t* __restrict__ offsets, uint32_t* __restrict__ perm) {
    const uint32_t M = *m_dev;
    if (M > capacity) return;        // the frame did not fit: the host sees the count and redoes it with larger buffers
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) perm[offsets[ray[i]] + idx[i]] = i;
}

__global__ void __launch_bounds__(kBlock) k_composite_records(const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ perm,
                                                              const float* __restrict__ w, const float* __restrict__ cd,
                                                              const float* __restrict__ cs, const float* __restrict__ ws, uint32_t N,
                                                              float intensity, float bg, float* __restrict__ image,
                                                              float* __restrict__ diffuse, float* __restrict__ specular,
                                                              const uint32_t* __restrict__ m_dev, uint32_t capacity, uint32_t tile_w) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    // the geometry pass appended the records block by block; with 8x8-pixel tiles as blocks (tile_w: the image width) a wave takes
    // the rays of one tile here too, so that what its lanes read through perm[] is one contiguous run of records again
    if (tile_w) { const uint32_t blk = r >> 6, lane = r & 63u, tiles_x = tile_w >> 3; r = ((blk / tiles_x) * 8u + (lane >> 3)) * tile_w + (blk % tiles_x) * 8u + (lane & 7u); }
    if (r >= N || *m_dev > capacity) return;
    float ar = 0, ag = 0, ab = 0, d0 = 0, d1 = 0, d2 = 0, s0 = 0, s1 = 0, s2 = 0;
    for (uint32_t j = offsets[r]; j < offsets[r + 1]; ++j) {
        const size_t i = perm[j];
        const float wi = w[i];
        if (wi == 0) continue;          // contributes w * c = 0 whatever c is -- and was not shaded when the records were gathered by weight
        const float c0 = cd[3 * i], c1 = cd[3 * i + 1], c2 = cd[3 * i + 2];
        const float e0 = cs[3 * i], e1 = cs[3 * i + 1], e2 = cs[3 * i + 2];
        ar += wi * ((c0 + e0) * intensity); ag += wi * ((c1 + e1) * intensity); ab += wi * ((c2 + e2) * intensity);
        d0 += wi * c0; d1 += wi * c1; d2 += wi * c2;
        s0 += wi * e0; s1 += wi * e1; s2 += wi * e2;
    }
    const float rest = 1 - ws[r];
    image[3 * (size_t)r] = ar + rest * bg; image[3 * (size_t)r + 1] = ag + rest * bg; image[3 * (size_t)r + 2] = ab + rest * bg;
    if (diffuse) { diffuse[3 * (size_t)r] = d0; diffuse[3 * (size_t)r + 1] = d1; diffuse[3 * (size_t)r + 2] = d2; }
    if (specular) { specular[3 * (size_t)r] = s0; specular[3 * (size_t)r + 1] = s1; specular[3 * (size_t)r + 2] = s2; }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
}  // namespace

namespace envidr {
int device_cu_count() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}
}  // namespace envidr

extern "C" {

uint32_t envidr_packed_weight_floats(int k_order, uint32_t k_in, uint32_t m_out) {
    return packed_weight_floats(k_order ? kTileOrder : kLaneOrder, k_in, m_out);
}
uint32_t envidr_packed_layer_floats(int k_order, uint32_t k_in, uint32_t m_out, int with_bias) {
    if (k_order == 2) return packed_weight_floats(kTileOrder, k_in, 16, with_bias != 0);     // one fragment per reduction step
    return packed_weight_floats((k_order == 1 || k_order == 3) ? kTileOrder : kLaneOrder, k_in, m_out, with_bias != 0);
}
int envidr_pack_layer(const float* W_host, const float* bias_host, uint32_t m_out, uint32_t k_in, int transpose, int k_order,
                      float* dst_host) {
    ENVIDR_REQUIRE(W_host && dst_host && m_out && k_in, "pack_layer: null pointer or empty layer");
    ENVIDR_REQUIRE(!(bias_host && transpose), "pack_layer: a transposed (gradient) layer carries no bias");
    ENVIDR_REQUIRE(k_order >= 0 && k_order <= 4, "pack_layer: k_order must be 0 (lane), 1 (tile), 2 (tile order, at most 16 outputs), 3 / 4 (tile / lane order, tail tile-major)");
    if (k_order == 2) {
        ENVIDR_REQUIRE(m_out <= 16 && !transpose, "pack_layer: k_order 2 packs a layer of at most 16 outputs, not transposed");
        pack_linear16(W_host, m_out, k_in, dst_host, bias_host);
        return ENVIDR_OK;
    }
    const KOrder order = (k_order == 1 || k_order == 3) ? kTileOrder : kLaneOrder;
    pack_linear(W_host, m_out, k_in, transpose != 0, order, dst_host, bias_host);
    if (k_order >= 3) {
        ENVIDR_REQUIRE(!transpose, "pack_layer: k_order 3 / 4 is for forward layers");
        const uint32_t mt = round_up(m_out, 32) / 32;
        retile_tail(dst_host + (bias_host ? (size_t)mt * 64 : 0), steps_for(order, k_in), mt);
    }
    return ENVIDR_OK;
}
uint32_t envidr_packed_rowvec_floats(uint32_t m_out) { return packed_bias_floats(m_out); }

int envidr_pack_linear(const float* W_host, uint32_t m_out, uint32_t k_in, int transpose, int k_order, float* dst_host) {
    ENVIDR_REQUIRE(W_host && dst_host && m_out && k_in, "pack_linear: null pointer or empty layer");
    pack_linear(W_host, m_out, k_in, transpose != 0, k_order ? kTileOrder : kLaneOrder, dst_host);
    return ENVIDR_OK;
}
int envidr_pack_rowvec(const float* v_host, uint32_t m_out, float* dst_host) {
    ENVIDR_REQUIRE(v_host && dst_host && m_out, "pack_rowvec: null pointer or empty vector");
    pack_rowvec(v_host, m_out, dst_host);
    return ENVIDR_OK;
}

uint64_t envidr_render_scratch_bytes(uint32_t N) { return (uint64_t)kScratchCounterWords * 4 + (uint64_t)N * 8; }

int envidr_render_rays(const envidr_render_desc* d, const float* rays_o, const float* rays_d, uint32_t N,
                       const envidr_render_out* out, uint32_t* ray_counter, envidr_stream_t stream) {
    ENVIDR_REQUIRE(d && out, "render_rays: null descriptor");
    if (N == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(rays_o && rays_d && ray_counter, "render_rays: null ray pointers / counter");
    ENVIDR_REQUIRE(out->image && out->depth && out->weights_sum, "render_rays: image, depth and weights_sum are required");
    ENVIDR_REQUIRE(d->density_bitfield && d->hash_table, "render_rays: null bitfield / hash table");
    ENVIDR_REQUIRE(d->num_levels >= 1 && d->num_levels <= ENVIDR_MAX_LEVELS, "render_rays: num_levels %u not in [1,16]", d->num_levels);
    ENVIDR_REQUIRE(d->cascades >= 1 && d->grid_size >= 1 && d->max_steps >= 1, "render_rays: bad grid parameters");
    ENVIDR_REQUIRE(d->beta > 0, "render_rays: beta must be positive");
    ENVIDR_REQUIRE(d->sdf_blob && d->head_blob && d->sdf_w3_row0 && (d->env_blob || d->dir_sh_degree), "render_rays: null weight blob");
    ENVIDR_REQUIRE(d->num_levels == ENVIDR_MAX_LEVELS, "render_rays: the fused kernel is built for 16 hash levels");

    RenderArgs a;
    memset(&a, 0, sizeof(a));
    a.rays_o = rays_o; a.rays_d = rays_d; a.N = N;
    a.mk = make_march_consts(d->bound, d->dt_gamma, d->max_steps, d->cascades, d->grid_size, d->density_bitfield);
    a.box = make_aabb(d);
    a.min_near = d->min_near; a.T_thresh = d->T_thresh; a.density_scale = d->density_scale; a.bg = d->bg_color;
    a.max_samples = d->max_steps;
    a.table = d->hash_table;
    a.num_levels = d->num_levels;
    a.bound2 = 2 * d->bound;
    {
        const char* err = fill_hash_levels(d, a.lv);
        ENVIDR_REQUIRE(!err, "render_rays: %s", err);
    }
    a.sdf_blob = d->sdf_blob; a.env_blob = d->env_blob; a.head_blob = d->head_blob;
    a.sdf_w3r0 = d->sdf_w3_row0;
    a.beta = d->beta; a.inv_beta = 1 / d->beta;
    a.rough_bias = d->roughness_bias; a.rough_act_scale = d->roughness_act_scale; a.rough_scale = d->roughness_scale;
    a.kappa_diffuse = d->diffuse_kappa_inv; a.light_scale = d->light_intensity_scale; a.intensity_scale = d->intensity_scale;
    a.has_rot = d->has_env_rot;
    for (int i = 0; i < 9; ++i) a.rot[i] = d->env_rot[i];
    a.geometry_only = d->geometry_only;
    if (d->r_images && !d->geometry_only) {
        ENVIDR_REQUIRE(d->renv_blob && d->spec2_blob, "render_rays: r_images given without the renv / second specular blobs");
        ENVIDR_REQUIRE(d->dir_sh_degree == 0, "render_rays: the reflected-radiance branch belongs to the environment-MLP family");
        a.r_images = d->r_images; a.renv_blob = d->renv_blob; a.spec2_blob = d->spec2_blob;
        a.indir_rough_thresh = d->indir_roughness_thresh;
    }
    if (d->geometry_export) {
        const envidr_geometry_export* e = d->geometry_export;
        ENVIDR_REQUIRE(d->geometry_only, "render_rays: geometry_export requires geometry_only");
        ENVIDR_REQUIRE(e->counter && e->ray && e->idx && e->w && e->normal && e->geo_feat && e->roughness,
                       "render_rays: null pointer in geometry_export");
        a.ex_counter = e->counter; a.ex_capacity = e->capacity; a.ex_ray = e->ray; a.ex_idx = e->idx; a.ex_w = e->w;
        a.ex_normal = e->normal; a.ex_geo = e->geo_feat; a.ex_rough = e->roughness;
    }
    a.image = out->image; a.depth = out->depth; a.ws = out->weights_sum; a.normal = out->normal_image;
    a.diffuse = out->diffuse_image; a.specular = out->specular_image; a.roughness = out->roughness_image;
    a.stats = reinterpret_cast<unsigned long long*>(out->stats);
    a.ray_counter = ray_counter;

    hipStream_t s = as_stream(stream);
    // work-list scratch: the caller's (desc->scratch) or the library's own (grows monotonically)
    uint32_t* hit_ids; float* hit_t; uint32_t* counters;
    if (d->scratch) {
        ENVIDR_REQUIRE(d->scratch_bytes >= envidr_render_scratch_bytes(N), "render_rays: scratch of %llu bytes, need %llu",
                       (unsigned long long)d->scratch_bytes, (unsigned long long)envidr_render_scratch_bytes(N));
        counters = reinterpret_cast<uint32_t*>(d->scratch);
        hit_ids = counters + kScratchCounterWords;
        hit_t = reinterpret_cast<float*>(hit_ids + N);
    } else {
        // one set per (device, stream), under a lock: renders enqueued on different streams or devices never share a work
        // list (a render on the SAME stream may: stream order serialises them)
        struct WorkScratch { uint32_t* hit_ids = nullptr; float* hit_t = nullptr; uint32_t* counters = nullptr; uint32_t cap = 0; };
        static std::mutex g_mu;
        static std::map<std::pair<int, hipStream_t>, WorkScratch> g_sets;
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lock(g_mu);
        WorkScratch& w = g_sets[{dev, s}];
        if (N > w.cap || !w.counters) {
            if (w.hit_ids) (void)hipFree(w.hit_ids);
            if (w.hit_t) (void)hipFree(w.hit_t);
            w.hit_ids = nullptr; w.hit_t = nullptr;
            if (!w.counters && hipMalloc(&w.counters, kScratchCounterWords * 4) != hipSuccess) return check_launch("render_rays counters alloc");
            w.cap = N + N / 4;
            if (hipMalloc(&w.hit_ids, (size_t)w.cap * 4) != hipSuccess || hipMalloc(&w.hit_t, (size_t)w.cap * 4) != hipSuccess) {
                w.hit_ids = nullptr; w.hit_t = nullptr; w.cap = 0;
                return check_launch("render_rays work-list alloc");
            }
        }
        hit_ids = w.hit_ids; hit_t = w.hit_t; counters = w.counters;
    }
    (void)ray_counter;   // kept in the ABI for callers that manage their own queue word; the library uses its own pair
    a.ray_counter = counters;
    a.hit_ids = hit_ids;
    a.hit_t = hit_t;
    a.ray_cost = d->ray_cost;
    if (hipMemsetAsync(counters, 0, kScratchCounterWords * 4, s) != hipSuccess) return check_launch("render_rays memset");
    hipLaunchKernelGGL(k_first_hit, dim3(ceil_div(N, kBlock)), dim3(kBlock), 0, s, a, hit_ids, hit_t, counters);
    {
        const int rc = check_launch("k_first_hit");
        if (rc) return rc;
    }
    if (a.ray_cost) {
        hipLaunchKernelGGL(k_order_hits, dim3(ceil_div(N, kBlock)), dim3(kBlock), 0, s, a, hit_ids, hit_t, counters);
        const int rc = check_launch("k_order_hits");
        if (rc) return rc;
    }

    // persistent grid: one 4-wave workgroup per CU (one wave per SIMD; the kernel uses the full
    // 512-register budget so exactly one wave fits a SIMD), fewer when the batch is small
    const uint32_t waves_per_block = kBlockThreads / 64;
    const uint32_t blocks = std::min((uint32_t)device_cu_count() * (4 / waves_per_block), ceil_div(N, kBlockThreads));
    const dim3 grid(blocks), block(kBlockThreads);
#define ENVIDR_LAUNCH(DEG, HT, SH) hipLaunchKernelGGL((k_render_persistent<DEG, HT, SH>), grid, block, 0, s, a)
    if (d->geometry_only) hipLaunchKernelGGL((k_render_persistent<4, 4, 0, true>), grid, block, 0, s, a);
    else if (d->dir_sh_degree == 4) ENVIDR_LAUNCH(4, 0, 4);
    else if (d->dir_sh_degree != 0) {
        set_error("render_rays: unsupported dir_sh_degree=%u (no-environment family is built for SH degree 4)", d->dir_sh_degree);
        return ENVIDR_EINVAL;
    }
    else if (d->ide_degree == 5 && d->env_hidden == 256) ENVIDR_LAUNCH(5, 8, 0);
    else if (d->ide_degree == 4 && d->env_hidden == 160) ENVIDR_LAUNCH(4, 5, 0);
    else if (d->ide_degree == 5 && d->env_hidden == 128) ENVIDR_LAUNCH(5, 4, 0);
    else if (d->ide_degree == 4 && d->env_hidden == 128) ENVIDR_LAUNCH(4, 4, 0);
    else {
        set_error("render_rays: unsupported (ide_degree=%u, env_hidden=%u); built variants: (5,256) (4,160) (5,128) (4,128)",
                  d->ide_degree, d->env_hidden);
        return ENVIDR_EINVAL;
    }
#undef ENVIDR_LAUNCH
    return check_launch("k_render_persistent");
}

// The counter is zeroed by a memset enqueued in front of the kernel that counts on it.  Two host threads enqueueing on the SAME stream
// must not interleave those pairs (memset, memset, kernel, kernel: the second kernel would start past M and shade nothing): the pair
// is enqueued under this lock.  (Enqueue only -- microseconds; different streams have different counters and only share the lock.)
static std::mutex g_work_counter_enqueue;

// one zero-initialised work counter per (device, stream): launches on different streams never share it
static uint32_t* shade_work_counter(hipStream_t s) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, uint32_t*> pool;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    uint32_t*& p = pool[{dev, s}];
    if (!p && hipMalloc(reinterpret_cast<void**>(&p), 256) != hipSuccess) p = nullptr;
    return p;
}

static int launch_shade(const envidr_render_desc* d, ShadeArgs& a, envidr_stream_t stream, const char* who) {
    ENVIDR_REQUIRE(d->head_blob && (d->env_blob || d->dir_sh_degree), "%s: null weight blob", who);
    a.env_blob = d->env_blob; a.head_blob = d->head_blob;
    a.kappa_diffuse = d->diffuse_kappa_inv; a.light_scale = d->light_intensity_scale;
    a.has_rot = d->dir_sh_degree ? 0 : d->has_env_rot;
    for (int i = 0; i < 9; ++i) a.rot[i] = d->env_rot[i];
    hipStream_t s = as_stream(stream);
    const bool renv = a.r_images != nullptr;
    // the heads-only instantiations (no environment MLP in the kernel: the SH family, and the split-precision mode whose
    // environment features are computed beforehand) need 216 registers: two waves per SIMD, which they want -- they are
    // latency-, not MFMA-bound
    const bool heads_only = d->dir_sh_degree != 0 || (d->env_split_blob != nullptr && !renv);
    const uint32_t waves_per_block = kBlockThreads / 64;
    const uint32_t blocks = std::min((uint32_t)device_cu_count() * ((heads_only ? 8 : 4) / waves_per_block), ceil_div(a.M, kBlockThreads));
    const dim3 grid(blocks), block(kBlockThreads);
    a.work = shade_work_counter(s);
    std::lock_guard<std::mutex> enqueue_lock(g_work_counter_enqueue);
    if (a.work && hipMemsetAsync(a.work, 0, sizeof(uint32_t), s) != hipSuccess) return check_launch("shade work counter");
    // split-precision mode: the environment features come from the fp16-pair kernel (shade_split.hip), the heads stay fp32
    const bool split = d->env_split_blob != nullptr && !d->dir_sh_degree;
    if (split) {
        ENVIDR_REQUIRE(!renv, "%s: split precision does not cover the reflected-radiance branch", who);
        a.env_pre = d->env_features;
        const int rc = d->env_split_form == 1 ? launch_env_split2(d, a, s, who) : launch_env_split(d, a, s, who);
        if (rc) return rc;
    }
#define ENVIDR_LAUNCH(DEG, HT) do { if (renv) hipLaunchKernelGGL((k_shade_samples<DEG, HT, 0, true>), grid, block, 0, s, a); \
                                    else if (split) hipLaunchKernelGGL((k_shade_samples<DEG, HT, 0, false, true>), grid, block, 0, s, a); \
                                    else hipLaunchKernelGGL((k_shade_samples<DEG, HT>), grid, block, 0, s, a); } while (0)
    if (d->dir_sh_degree == 4) {
        ENVIDR_REQUIRE(!renv, "%s: the reflected-radiance branch belongs to the environment-MLP family", who);
        ENVIDR_REQUIRE(a.ray_ids || a.dirs, "%s: the no-environment family needs view directions", who);
        hipLaunchKernelGGL((k_shade_samples<4, 0, 4>), grid, block, 0, s, a);
    } else if (d->dir_sh_degree != 0) {
        set_error("%s: unsupported dir_sh_degree=%u (no-environment family is built for SH degree 4)", who, d->dir_sh_degree);
        return ENVIDR_EINVAL;
    }
    else if (d->ide_degree == 5 && d->env_hidden == 256) ENVIDR_LAUNCH(5, 8);
    else if (d->ide_degree == 4 && d->env_hidden == 160) ENVIDR_LAUNCH(4, 5);
    else if (d->ide_degree == 5 && d->env_hidden == 128) ENVIDR_LAUNCH(5, 4);
    else if (d->ide_degree == 4 && d->env_hidden == 128) ENVIDR_LAUNCH(4, 4);
    else {
        set_error("%s: unsupported (ide_degree=%u, env_hidden=%u); built variants: (5,256) (4,160) (5,128) (4,128)", who,
                  d->ide_degree, d->env_hidden);
        return ENVIDR_EINVAL;
    }
#undef ENVIDR_LAUNCH
    return check_launch("k_shade_samples");
}

int envidr_shade_samples(const envidr_render_desc* d, const float* normals, const float* dirs, const float* geo_feat,
                         uint32_t geo_feat_stride, const float* roughness, uint32_t roughness_stride, uint32_t M,
                         float* c_diffuse, float* c_specular, envidr_stream_t stream) {
    ENVIDR_REQUIRE(d, "shade_samples: null descriptor");
    if (M == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(normals && dirs && geo_feat && roughness && c_diffuse && c_specular, "shade_samples: null pointer");
    ENVIDR_REQUIRE((geo_feat_stride == 0 || geo_feat_stride == 12) && roughness_stride <= 1,
                   "shade_samples: geo_feat_stride must be 0 or 12 and roughness_stride 0 or 1");
    ShadeArgs a;
    memset(&a, 0, sizeof(a));
    a.normals = normals; a.dirs = dirs; a.geo_feat = geo_feat; a.roughness = roughness;
    a.geo_stride = geo_feat_stride; a.rough_stride = roughness_stride; a.M = M;
    a.c_diffuse = c_diffuse; a.c_specular = c_specular;
    return launch_shade(d, a, stream, "shade_samples");
}

int envidr_shade_records(const envidr_render_desc* d, const envidr_geometry_export* rec, const float* rays_d, float* c_diffuse,
                         float* c_specular, envidr_stream_t stream) {
    ENVIDR_REQUIRE(d && rec, "shade_records: null descriptor / records");
    if (rec->capacity == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(rec->counter && rec->ray && rec->normal && rec->geo_feat && rec->roughness && rays_d && c_diffuse && c_specular,
                   "shade_records: null pointer");
    ENVIDR_REQUIRE(!rec->shade_list || rec->w, "shade_records: shade_list needs the records' weights");
    ShadeArgs a;
    memset(&a, 0, sizeof(a));
    a.normals = rec->normal; a.geo_feat = rec->geo_feat; a.roughness = rec->roughness;
    a.geo_stride = 12; a.rough_stride = 1; a.M = rec->capacity;
    a.ray_ids = rec->ray; a.rays_d = rays_d; a.m_dev = rec->counter;
    a.slot = rec->slot;
    if (d->r_images) {
        // third pass of indirect rendering (network.py:612-659): per-ray reflected radiance, indexed by the records' ray ids
        ENVIDR_REQUIRE(d->renv_blob && d->spec2_blob, "shade_records: r_images given without the renv / second specular blobs");
        ENVIDR_REQUIRE(rec->blend, "shade_records: the reflected-radiance branch needs the records' blend logits");
        a.r_images = d->r_images; a.blend = rec->blend; a.renv_blob = d->renv_blob; a.spec2_blob = d->spec2_blob;
        a.rough_scale = d->roughness_scale; a.indir_rough_thresh = d->indir_roughness_thresh;
    }
    a.c_diffuse = c_diffuse; a.c_specular = c_specular;
    if (rec->shade_list && !(d->env_split_blob && !d->dir_sh_degree && d->env_split_form != 1)) {
        hipStream_t s = as_stream(stream);
        const uint32_t n_blocks = ceil_div(rec->capacity, kGatherChunk);
        uint32_t* counts = rec->shade_list + 1 + rec->capacity;          // behind the list (envidr_render.h: capacity + 1 + capacity / 1024 + 1 words)
        hipLaunchKernelGGL(k_count_weighted_records, dim3(n_blocks), dim3(kBlock), 0, s, rec->w, rec->counter, rec->capacity, counts);
        hipLaunchKernelGGL(k_scan_record_counts, dim3(1), dim3(1024), 0, s, counts, n_blocks, rec->shade_list);
        hipLaunchKernelGGL(k_scatter_weighted_records, dim3(n_blocks), dim3(kBlock), 0, s, rec->w, rec->counter, rec->capacity, counts, rec->shade_list);
        const int rc = check_launch("k_scatter_weighted_records");
        if (rc) return rc;
        a.list = rec->shade_list + 1;
        a.m_all = rec->counter;
        a.m_dev = rec->shade_list;
    }
    return launch_shade(d, a, stream, "shade_records");
}

int envidr_env_mlp_forward(const float* env_blob, uint32_t in_dim, uint32_t hidden, const float* x, uint32_t M, float* y, envidr_stream_t stream) {
    if (M == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(env_blob && x && y, "env_mlp_forward: null pointer");
    hipStream_t s = as_stream(stream);
    uint32_t* work = shade_work_counter(s);
    ENVIDR_REQUIRE(work, "env_mlp_forward: no memory for the work counter");
    std::lock_guard<std::mutex> enqueue_lock(g_work_counter_enqueue);
    if (hipMemsetAsync(work, 0, sizeof(uint32_t), s) != hipSuccess) return check_launch("env_mlp_forward work counter");
    const uint32_t blocks = std::min((uint32_t)device_cu_count() * 4u, ceil_div(M, 64u));
    const uint32_t aligned = aligned16(x) && aligned16(y) ? 1u : 0u;
#define ENVIDR_MLP(DEG, HT) hipLaunchKernelGGL((k_env_mlp<DEG, HT>), dim3(blocks), dim3(kBlockThreads), 0, s, env_blob, x, y, M, work, aligned)
    if (in_dim == 72 && hidden == 256) ENVIDR_MLP(5, 8);
    else if (in_dim == 38 && hidden == 160) ENVIDR_MLP(4, 5);
    else if (in_dim == 72 && hidden == 128) ENVIDR_MLP(5, 4);
    else if (in_dim == 38 && hidden == 128) ENVIDR_MLP(4, 4);
    else {
        set_error("env_mlp_forward: unsupported (in_dim=%u, hidden=%u); built variants: (72,256) (38,160) (72,128) (38,128)", in_dim, hidden);
        return ENVIDR_EINVAL;
    }
#undef ENVIDR_MLP
    return check_launch("k_env_mlp");
}

int envidr_composite_records(const envidr_geometry_export* rec, const uint32_t* offsets, uint32_t* perm, const float* c_diffuse,
                             const float* c_specular, const float* weights_sum, uint32_t N, float intensity_scale, float bg_color,
                             float* image, float* diffuse_image, float* specular_image, envidr_stream_t stream) {
    if (N == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(rec && rec->counter && rec->ray && rec->idx && rec->w && offsets && perm && c_diffuse && c_specular && weights_sum && image,
                   "composite_records: null pointer");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_place_records, dim3(std::min(ceil_div(rec->capacity ? rec->capacity : 1u, kBlock), 16384u)), dim3(kBlock), 0, s,
                       rec->ray, rec->idx, rec->counter, rec->capacity, offsets, perm);
    {
        const int rc = check_launch("k_place_records");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_composite_records, dim3(ceil_div(N, kBlock)), dim3(kBlock), 0, s, offsets, perm, rec->w, c_diffuse, c_specular,
                       weights_sum, N, intensity_scale, bg_color, image, diffuse_image, specular_image, rec->counter, rec->capacity,
                       (rec->image_width && rec->image_width % 8u == 0 && N % rec->image_width == 0 && (N / rec->image_width) % 8u == 0) ? rec->image_width : 0u);
    return check_launch("k_composite_records");
}

int envidr_composite_shaded(const uint32_t* offsets, const float* w, const float* c_diffuse, const float* c_specular,
                            const float* weights_sum, uint32_t N, float intensity_scale, float bg_color, float* image,
                            float* diffuse_image, float* specular_image, envidr_stream_t stream) {
    if (N == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(offsets && w && c_diffuse && c_specular && weights_sum && image, "composite_shaded: null pointer");
    hipLaunchKernelGGL(k_composite_shaded, dim3(ceil_div(N, kBlock)), dim3(kBlock), 0, as_stream(stream), offsets, w, c_diffuse,
                       c_specular, weights_sum, N, intensity_scale, bg_color, image, diffuse_image, specular_image);
    return check_launch("k_composite_shaded");
}

}  // extern "C"
