// Geometry pipeline of the fused renderer for gfx950 (include/envidr_render.h, envidr_geometry_*).
//
// The geometry half of a frame -- march -> hash grid -> SDF network forward + input gradient -> density, normal,
// geometry feature, roughness -> compositing weights -- as a device-driven sequence of two kinds of launches:
//
//   k_geo_rays   one wave per block of 64 neighbouring rays, lane = ray: composites the samples its rays got evaluated
//                in the previous round (reference recurrence, raymarching.cu:996-1030), finishes rays that terminated
//                or left the scene, and marches the next chunk of samples of the others (march_core.hip.h, the
//                standalone operator's bit-exact code) into a compact sample list, laid out sample-major per block so
//                that every access of the wave is contiguous (see "Memory layout of the rounds").  The first chunk is
//                sized from the caller's per-ray hint when there is one, else 16; later chunks are sized per ray from
//                what the compositor has just seen (transmittance left and the last opacity: see "chunk prediction").
//                No host round trip: counts and the list of live blocks stay on the device, the host enqueues a fixed
//                number of rounds.
//   k_geo_eval32 one lane per (sample, half of the hash levels) of the round's list, nothing per ray: 16-level hash
//                gathers with analytic Jacobian, SDF network forward and backward on the matrix cores, per-sample
//                geometry terms.  32 samples per wave, two waves per SIMD (one in its gather / interpolation phase
//                while the other owns the matrix pipe); the SDF weights are read from HBM once per workgroup and stay
//                in LDS, and so does the Jacobian of a batch until the backward pass has produced the feature gradients.
//                (k_geo_eval16, geo_eval16.hip.h, is the same evaluation on 16-column MFMAs: a tested, selectable
//                alternative.  Earlier forms and their A/B switches live in tools/geo/variants.py as source patches.)
//
// Sample positions are those of the reference loop with one sample per ray per iteration (the marcher is resumed
// from the composited ray time after every sample, which does not depend on the densities), i.e. exactly the samples
// of envidr_render_rays; the per-sample arithmetic is fused_common.hip.h's, shared with that kernel.
#include "fused_common.hip.h"
#include "hash_lean.hip.h"

#include <algorithm>
#include <utility>

using namespace envidr;

namespace {

// ---- SDF weights resident in LDS ----------------------------------------------------------------------------
// Fragment I of the SDF blob (mlp_mfma.hip.h: [fragment][64 lanes]) is one conflict-free ds_read_b32.  The reads run
// PF fragments ahead of the MFMAs that consume them through a small register ring, pinned with one sched_barrier per
// reduction step (pipe_steps): left to itself the scheduler hoists dozens of these reads ahead of their use and the
// kernel no longer fits two waves per SIMD.  Every pass streams the same blob, so the ring wraps around to fragment 0;
// end_pass pads a pass to a multiple of PF so that fragment j always sits in slot j % PF.
template <int PF>
struct WeightLdsRing {
    const float* frag;     // LDS base of the blob + lane
    float ring[PF];
    __device__ __forceinline__ void start(const float* lds_blob_lane) {
        frag = lds_blob_lane;
#pragma unroll
        for (int i = 0; i < PF; ++i) ring[i] = frag[i * 64];
    }
    __device__ __forceinline__ void begin_pass(const float*, uint32_t, const float*, uint32_t) {}
    template <int I, int FRAGS>
    __device__ __forceinline__ float take() {
        static_assert(FRAGS % PF == 0, "pass length must be a multiple of the ring depth");
        const float v = ring[I % PF];
        ring[I % PF] = frag[((I + PF) % FRAGS) * 64];
        return v;
    }
    template <int FRAGS, int I = FRAGS>
    __device__ __forceinline__ void end_pass() {
        if constexpr (I % PF != 0) {
            constexpr int kPadded = (FRAGS + PF - 1) / PF * PF;
            ring[I % PF] = frag[((I + PF) % kPadded) * 64];      // the next pass's fragment that belongs in this slot
            end_pass<FRAGS, I + 1>();
        }
    }
};
}  // namespace
namespace envidr {
template <int PF> struct is_weight_ring<WeightLdsRing<PF>> { static constexpr bool value = true; };
}
namespace {


constexpr int kGeoRing = 8;                         // LDS weight fragments read ahead of the MFMAs that consume them
constexpr int kGeoAhead = 2;                        // hash levels whose corner gathers are in flight ahead of the one being interpolated
constexpr int kSdfBlobFloats = kSdfFrags * 64;
constexpr int kW3RowFloats = 64;                    // packed row vector of W3[0, :] (two tiles x 32)

struct GeoEvalArgs {
    const float* xyz;           // [cap,3] sample positions
    const float* dt;            // [cap]   step the compositor uses for alpha
    const uint32_t* range;      // device {begin, count} of the samples to evaluate, or null: [0, M)
    const uint32_t* head;       // pipeline mode: evaluate [*begin_io, *head) and leave begin_io[1] = *head for the next round
    uint32_t* begin_io;
    uint32_t M;
    // hash grid
    const float* table;
    uint32_t table_bytes;
    LeanLevel lv[kLevels];
    float bound, bound2;
    // SDF network
    const float* sdf_blob;
    const float* sdf_w3r0;
    const float* sdf_e16_blob;  // k_geo_eval16's packing of the same network (geo_eval16.hip.h), or null
    float inv_beta, beta, density_scale;
    float rough_bias, rough_act_scale, rough_scale;
    // per-sample outputs (any may be null)
    float* alpha;               // [cap]    1 - exp(-sigma dt)
    float* sigma;               // [cap]
    float* normal;              // [cap,3]  unit
    float* geo;                 // [cap,12] unit
    float* rough;               // [cap]
    float* blend;               // [cap]    raw output 14 of the SDF network (learn_indir_blend logit)
    // test-only exports of the PROBE instantiation (envidr_geometry_probe): what the hash section computed
    float* probe_feat;          // [M,32]   hash features as they enter the SDF network (level-major, 2 channels)
    uint32_t* probe_rows;       // [M,16,8] table row (within its level) of corner bx | by << 1 | bz << 2
    float* probe_raw;           // [M,16]   raw outputs of the last SDF layer (row 0 = sdf)
    float* probe_grad;          // [M,3]    d sdf / d xyz before normalisation
};

// compiler-level ordering of this wave's LDS traffic between two phases (a wave's DS operations execute in order;
// nothing is emitted for wavefront scope)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// =====================================================================================================================
// k_geo_eval32: the same evaluation with 32 samples per wave and the 16 hash levels split between the two lane halves
// =====================================================================================================================
// Lane (s, h) -- sample s = lane & 31 of the batch, half h = lane >> 5 -- evaluates the eight levels 2 i + h.  What this
// buys: the Jacobian a wave parks is 48 values per lane (12 KiB per wave), so EIGHT waves -- two per SIMD -- fit beside the
// LDS-resident weights, and one wave's gather phase (fabric-bound) overlaps its SIMD partner's matrix-core phase; the
// matrix-core section of a wave is one 32-sample group; all 64 lanes gather.  The half-wave exchange the MFMA operand
// packing needs anyway (v_permlane32_swap) also brings a level's two channels together for the first layer and hands
// every lane the feature gradients of its own levels afterwards.
constexpr int kE32Waves = 8;
constexpr int kE32Threads = kE32Waves * 64;
constexpr int kE32Steps = kLevels / 2;            // level pairs: step i covers level 2 i (lower lanes) and 2 i + 1 (upper lanes)

// PROBE: the test-only instantiation behind envidr_geometry_probe -- the same statements, plus stores of what the hash section
// computed (features, corner rows) and of the raw network outputs / the unnormalised gradient
template <bool PROBE>
__global__ void __launch_bounds__(kE32Threads, 2) k_geo_eval32(const GeoEvalArgs a) {
    __shared__ __attribute__((aligned(16))) float s_w[kSdfBlobFloats + kW3RowFloats];
    __shared__ __attribute__((aligned(16))) LeanLevel s_lv[kLevels];
    __shared__ float s_jac[kE32Waves * kE32Steps * 6 * 64];
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t half = lane >> 5, sl = lane & 31u;
    uint32_t begin = 0, count = a.M;
    if (a.range) { begin = __builtin_amdgcn_readfirstlane(a.range[0]); count = min(__builtin_amdgcn_readfirstlane(a.range[1]), a.M - min(begin, a.M)); }
    else if (a.head) {
        begin = __builtin_amdgcn_readfirstlane(a.begin_io[0]);
        const uint32_t end = min(__builtin_amdgcn_readfirstlane(*a.head), a.M);
        count = end - min(begin, end);
        if (blockIdx.x == 0 && threadIdx.x == 0) a.begin_io[1] = end;
    }
    const uint32_t batches = (count + 31u) / 32u;
    // XCD-aware batch order: workgroup b runs on XCD b % 8 (observed; speed only), and each XCD takes one contiguous
    // eighth of the sample list, so image-space neighbours -- whose gathers share table rows -- meet in the same L2
    const uint32_t xcd = blockIdx.x & 7u, in_xcd = blockIdx.x >> 3, per_xcd_blocks = (gridDim.x + 7u - xcd) >> 3;
    const uint32_t share = (batches + 7u) / 8u;
    const uint32_t b_lo = min(xcd * share, batches), b_hi = min(b_lo + share, batches);
    if (b_lo + in_xcd * kE32Waves >= b_hi) return;          // nothing for this workgroup (most rounds of a frame are empty)
    {
        const float4* src = reinterpret_cast<const float4*>(a.sdf_blob);
        float4* dst = reinterpret_cast<float4*>(s_w);
        for (uint32_t i = threadIdx.x; i < kSdfBlobFloats / 4; i += kE32Threads) dst[i] = src[i];
        if (threadIdx.x < kW3RowFloats) s_w[kSdfBlobFloats + threadIdx.x] = a.sdf_w3r0[threadIdx.x];
        if (threadIdx.x < kLevels) s_lv[threadIdx.x] = a.lv[threadIdx.x];
    }
    __syncthreads();
    WeightLdsRing<kGeoRing> wp;
    wp.start(s_w + lane);
    const float* w3row = s_w + kSdfBlobFloats + half * 16;
    f32x2* jac_col = reinterpret_cast<f32x2*>(s_jac + wave * (kE32Steps * 6 * 64)) + lane;      // [step * 3 + axis][lane] of channel pairs
    constexpr int kSdfN = (kSdfFrags + kGeoRing - 1) / kGeoRing * kGeoRing;
    constexpr int kAhead = kGeoAhead;
    const __amdgpu_buffer_rsrc_t table = table_rsrc(a.table, a.table_bytes);
    const uint32_t stride = per_xcd_blocks * kE32Waves;

    for (uint32_t b = b_lo + in_xcd * kE32Waves + wave; b < b_hi; b += stride) {
        const uint32_t sidx = b * 32u + sl;
        const bool on = sidx < count;
        const size_t slot = (size_t)begin + (on ? sidx : 0u);
        float xc[3];
        bool inside;
        {
            typedef float f32x3 __attribute__((ext_vector_type(3), aligned(4)));   // 12-byte stride arrays: NOT 16-byte aligned
            f32x3 pv = {0.0f, 0.0f, 0.0f};
            if (on) pv = *reinterpret_cast<const f32x3*>(a.xyz + 3 * slot);
            const float x01[3] = {(pv[0] + a.bound) / a.bound2, (pv[1] + a.bound) / a.bound2, (pv[2] + a.bound) / a.bound2};
            inside = x01[0] >= 0 && x01[0] <= 1 && x01[1] >= 0 && x01[1] <= 1 && x01[2] >= 0 && x01[2] <= 1;
#pragma unroll
            for (int d = 0; d < 3; ++d) xc[d] = inside ? x01[d] : 0.5f;
        }

        // ================= phase 1: this lane's eight levels: values + Jacobian -> LDS ============================
        float f0[kE32Steps], f1[kE32Steps];
        {
            LeanStage st[kAhead + 1];
            LeanLevel lvs[kAhead + 1];
            auto prep = [&](int i, LeanStage& stg, LeanLevel& lv) {
                lv = s_lv[2 * i + half];                                       // per-lane level constants (two distinct rows: broadcast)
                uint32_t rows[8];
                lean_prepare<0, true>(lv, table, xc, stg, PROBE ? rows : nullptr);
                if constexpr (PROBE) {
                    if (on && a.probe_rows) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) a.probe_rows[(slot * kLevels + (2 * i + half)) * 8 + c] = rows[c];
                    }
                }
            };
            [&]<int... I>(std::integer_sequence<int, I...>) { (prep(I, st[I], lvs[I]), ...); }(std::make_integer_sequence<int, kAhead>{});
            __builtin_amdgcn_sched_barrier(0);
            auto step = [&](auto ic, LeanStage& now, LeanLevel& lvnow, LeanStage& ahead, LeanLevel& lvahead) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i + kAhead < kE32Steps) prep(i + kAhead, ahead, lvahead);
                __builtin_amdgcn_sched_barrier(0);
                f32x2 o, g[3];
                lean_finish(now, inside ? lvnow.on : 0.0f, o, g);
                f0[i] = o[0]; f1[i] = o[1];
                if constexpr (PROBE) {
                    if (on && a.probe_feat) {
                        a.probe_feat[slot * (2 * kLevels) + 2 * (2 * i + half)] = o[0];
                        a.probe_feat[slot * (2 * kLevels) + 2 * (2 * i + half) + 1] = o[1];
                    }
                }
#pragma unroll
                for (int d = 0; d < 3; ++d) jac_col[(i * 3 + d) * 64] = g[d];          // one 8-byte LDS write per axis
            };
            [&]<int... I>(std::integer_sequence<int, I...>) {
                (step(std::integral_constant<int, I>{}, st[I % (kAhead + 1)], lvs[I % (kAhead + 1)], st[(I + kAhead) % (kAhead + 1)],
                      lvs[(I + kAhead) % (kAhead + 1)]), ...);
            }(std::make_integer_sequence<int, kE32Steps>{});
        }
        // first-layer operands: step q = level q takes (channel 0 | channel 1) of the 32 samples in the (lower | upper) lanes
        float in[kLevels];
#pragma unroll
        for (int i = 0; i < kE32Steps; ++i) {
            float u = f0[i], v = f1[i];
            swap_halves(u, v);               // u = [ch0 | ch1] of level 2 i,  v = [ch0 | ch1] of level 2 i + 1
            in[2 * i] = u; in[2 * i + 1] = v;
        }

        // ================= phase 2: SDF network forward + input gradient: one 32-sample group ===================
        // (the hidden pre-activations h1, h2 stay in their accumulator tiles until the backward pass has used their signs: a ReLU
        //  mask is then one compare + one select per value where it is needed; packing the signs into bit words and unpacking
        //  them again cost ~450 of the ~2900 vector instructions of a batch, and this kernel is issue-bound)
        f32x16 o3, gf[1];
        {
            f32x16 h1[2], h2[2];
            pipe_layer_from_lanes<kLevels, 2, kSdfW1, kSdfN>(wp, lane, in, h1);
            pipe_layer_from_tiles<2, 2, kSdfW2, kSdfN, true>(wp, lane, h1, h2);
            pipe_layer16_from_tiles<2, kSdfW3, kSdfN, true>(wp, lane, h2, o3);        // 64 -> 15 on 16-row MFMA blocks (k_order 2)
            f32x16 g2[2], g1[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) g2[t][r] = h2[t][r] > 0 ? w3row[t * 32 + r] : 0.0f;
            pipe_layer_from_tiles<2, 2, kSdfW2t, kSdfN, false, false>(wp, lane, g2, g1);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) g1[t][r] = h1[t][r] > 0 ? g1[t][r] : 0.0f;
            pipe_layer_from_tiles<2, 1, kSdfW1t, kSdfN, false, false>(wp, lane, g1, gf);
            wp.template end_pass<kSdfFrags>();
        }

        // ================= phase 3: every lane gets d sdf / d (its own levels' features); normal = J^T g ===================
        // register r of the gradient tile holds feature (r & 3) + 8 (r >> 2) + 4 h: per quad of registers one exchange per
        // channel hands both halves the gradients of their levels 2 i + h for i = 2 k and 2 k + 1
        float g0[kE32Steps], g1c[kE32Steps];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float u = gf[0][4 * k], v = gf[0][4 * k + 2];
            swap_halves(u, v);
            g0[2 * k] = u; g0[2 * k + 1] = v;
            float p = gf[0][4 * k + 1], q = gf[0][4 * k + 3];
            swap_halves(p, q);
            g1c[2 * k] = p; g1c[2 * k + 1] = q;
        }
        float part[3] = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < kE32Steps; ++i)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const f32x2 j = jac_col[(i * 3 + d) * 64];
                part[d] += g0[i] * j[0];
                part[d] += g1c[i] * j[1];
            }
        float nrm[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float lo = part[d], hi = part[d];
            swap_halves(lo, hi);             // lo = the lower half's sum in every lane, hi = the upper half's
            nrm[d] = (lo + hi) / a.bound2;                                              // even levels + odd levels; d x01 / d xyz
        }
        if constexpr (PROBE) {
            if (on && half == 0 && a.probe_grad) {
#pragma unroll
                for (int d = 0; d < 3; ++d) a.probe_grad[slot * 3 + d] = nrm[d];
            }
        }
        normalize_n<3>(nrm, 1e-10f);                                                    // renderer.py:192
        // raw outputs 0..15 of the last layer: rows 0-3, 8-11 in the lower lanes' registers 0..7, rows 4-7, 12-15 in the upper lanes'
        float h3[16];
        {
            // the 16-row blocks hold rows 4 Q + q of sample (lane & 15) [+ 16] in lane quarter Q; both lane halves hold the same 32
            // samples here, so "group A" and "group B" of the transpose are the same accumulator
            float lo[4], hi[4];
            fold16(o3, lo, hi);
            rows_to_lanes<4>(lo, hi, lo, hi, h3);
        }
        if constexpr (PROBE) {
            if (on && half == 0 && a.probe_raw) {
#pragma unroll
                for (int i = 0; i < 16; ++i) a.probe_raw[slot * 16 + i] = h3[i];
            }
        }
        const float sdf = h3[0];
        float geo[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) geo[i] = h3[1 + i];
        normalize_n<12>(geo, 1e-12f);                                                   // network.py:434-435
        const float rough = a.rough_act_scale * softplusf(h3[13] + a.rough_bias) * a.rough_scale;   // network.py:443-448
        // Laplace density (network.py:32-37): (1/beta) (0.5 + 0.5 sign(s) expm1(-|s| / beta))
        const float sgn = sdf > 0 ? 1.0f : (sdf < 0 ? -1.0f : 0.0f);
        const float sigma = a.inv_beta * (0.5f + 0.5f * sgn * expm1f(-fabsf(sdf) / a.beta)) * a.density_scale;
        if (on && half == 0) {
            if (a.alpha) a.alpha[slot] = 1.0f - expf(-sigma * a.dt[slot]);
            if (a.sigma) a.sigma[slot] = sigma;
            if (a.rough) a.rough[slot] = rough;
            if (a.blend) a.blend[slot] = h3[14];
            if (a.normal) {
                typedef float f32x3 __attribute__((ext_vector_type(3), aligned(4)));   // 12-byte stride arrays: NOT 16-byte aligned
                const f32x3 nv = {nrm[0], nrm[1], nrm[2]};
                *reinterpret_cast<f32x3*>(a.normal + 3 * slot) = nv;
            }
            if (a.geo) {
                float4* gp = reinterpret_cast<float4*>(a.geo + 12 * slot);
#pragma unroll
                for (int i = 0; i < 3; ++i) gp[i] = make_float4(geo[4 * i], geo[4 * i + 1], geo[4 * i + 2], geo[4 * i + 3]);
            }
        }
        wave_lds_sync();      // the parked Jacobian is rewritten by the next batch
    }
}

}  // namespace
#include "geo_eval16.hip.h"
namespace {

// =====================================================================================================================
// per-ray rounds
// =====================================================================================================================
// Memory layout of the rounds.  A *block* is the 64 rays 64 b .. 64 b + 63 -- one wave, lane = ray -- for the whole pass:
// rays, hints, per-ray outputs and the per-ray state (structure of arrays) are then read and written by consecutive lanes
// at consecutive addresses.  The chunk a block marches in a round occupies one contiguous run of sample slots, laid out
// SAMPLE-major: first every ray's sample 0, then every ray's sample 1 (of the rays that have one) ...:
//     slot(lane, c) = base + sum_{c' < c} popcount(M_c') + popcount(M_c & lanes below),   M_c = lanes with more than c samples
// so that at every step of the march (and of the compositing a round later, which rebuilds the same masks from the
// per-ray counts) the lanes of a wave touch one contiguous piece of every sample array.  With one run per ray instead,
// each lane access is its own 64-byte line request, and both per-ray kernels ran at the chip's line-request rate
// (~80 G lines/s: 0.8 and 1.0 ms per frame) rather than at anything to do with their arithmetic.
// Records are appended the same way (sample-major within the block), which also makes the shading kernel's reads through
// rec->slot and the final compositor's reads through perm[] nearly sequential.
// Blocks are never re-packed: the list a round consumes is the ids of the blocks that still have a live ray.
enum RayField : int { kFAccT = 0, kFWs, kFDepth, kFAn0, kFAn1, kFAn2, kFRough, kFTaken, kFChunk, kFAlloc, kStateFields };
struct RayState {
    float acc_t;             // composited ray time = where the marcher resumes (the reference re-derives it from the deltas)
    float ws, depth;
    float an[3];             // sum w * normal
    float arough;            // sum w * roughness
    uint32_t n_taken;        // samples composited so far
    uint32_t chunk_count;    // samples marched last round (0: the ray is finished); bit 31: the ray ran out of samples inside it
};
// (state field kFAlloc: the slots a ray was GIVEN for its chunk, >= chunk_count -- an over-estimating hint leaves zero-filled
//  slots -- kept for every lane of a live block: the sample-major slot layout of the block is defined by these counts)

// counters (device uint32[kGeoCounterWords]), zeroed by the host before every frame
constexpr uint32_t kCntSampleHead = 0, kCntRecordHead = 1, kCntOverflow = 2, kCntAlive = 8, kCntBegin = 40, kCntOcc = 72, kGeoCounterWords = 256;
// (words 80 .. : unused by the product, zeroed with the rest; the timer variant of tools/geo/build_variants.py accumulates there)
// kCntOcc + 0..2: max over occupied cells of (H - 1 - coordinate) (i.e. the minimum, as a maximum: the words start at zero),
// kCntOcc + 3..5: max coordinate; written by k_linearize_bitfield
constexpr uint32_t kMaxRounds = 30;     // counter words reserved per round-indexed array

struct GeoRayArgs {
    const float* rays_o; const float* rays_d;
    uint32_t N;
    MarchConsts mk;
    Aabb box;
    float min_near, T_thresh;
    uint32_t max_samples;
    uint32_t chunk;            // samples to march this round (round 0: per ray without a hint; later rounds: the most a ray may
                               // get, see predicted_chunk); 0: only composite (last round)
    uint32_t predict;          // rounds >= 1: size each ray's chunk from its compositing state instead of taking `chunk`
    uint32_t round;
    uint32_t* counters;
    const uint32_t* alive_in;  // rounds >= 1: ids of the blocks that still have a live ray
    uint32_t* alive_out;
    uint32_t* block_base;      // [blocks] first slot of the chunk the block marched last
    uint32_t* state;           // [kStateFields][n_pad] (floats and counts)
    uint32_t n_pad;
    // sample list
    float* xyz; float* dt; float* dd;
    uint32_t cap;
    const float* alpha; const float* normal; const float* rough;     // written by k_geo_eval for the previous chunk
    // records: one per composited sample
    uint32_t* rec_ray; uint32_t* rec_idx; float* rec_w; uint32_t* rec_slot;
    uint32_t rec_cap;
    // per-ray outputs
    float* depth; float* ws; float* nimg; float* rimg;
    uint16_t* ray_cost;
    const uint8_t* ray_mask;   // optional: rays whose byte is 0 are finished at once (round 0)
    const uint8_t* linear_grid; // one-cascade fast marcher (march_core.hip.h): the bitfield in linear cell order, or null
    uint32_t log2H;
    uint32_t occ_clip;          // ray_range(): clip rays to the box of the occupied cells
    uint32_t tile_w, tiles_x;   // tile_w != 0: the rays are a row-major image tile_w pixels wide and block b is the 8x8-pixel tile
                                // (b / tiles_x, b % tiles_x), lane l its pixel (l / 8, l % 8); else block b = rays 64 b .. 64 b + 63
};

// [near, far) of a ray: the slab test against the model's box (near_far: the reference's arithmetic), with `far` pulled in to
// where the ray leaves the bounding box of the OCCUPIED cells, grown by a cell on every side (a ray that misses that box
// gets an empty range).  Beyond that point the marcher can only skip empty cells until it runs out at `far`, so the
// samples are the same; what goes away is that walk -- ~200 cells for every ray that misses the object, the longest
// lanes of the first round.  Only with the linear bitfield copy (its kernel reduces the box) and a model box inside the cube.
__device__ __forceinline__ Aabb occupied_box(const GeoRayArgs& a) {       // wave-uniform: once per wave, ahead of the block loop
    Aabb ob = a.box;
    if (a.occ_clip) {
        const uint32_t H1 = (1u << a.log2H) - 1u;
        const float cs = 2.0f * a.mk.bound / (float)(H1 + 1u);
        uint32_t w[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) w[q] = a.counters[kCntOcc + q];            // six independent loads: one latency
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int lo = (int)H1 - (int)w[d], hi = (int)w[3 + d];
            ob.lo[d] = -a.mk.bound + (float)(lo - 1) * cs;
            ob.hi[d] = -a.mk.bound + (float)(hi + 2) * cs;
        }
    }
    return ob;
}
__device__ __forceinline__ void ray_range(const GeoRayArgs& a, const Aabb& ob, const RayGeom& rg, float& near, float& far) {
    near_far(rg, a.box, a.min_near, near, far);
    if (a.occ_clip) {
        float n2, f2;
        near_far(rg, ob, 0.0f, n2, f2);
        far = n2 == FLT_MAX ? 0.0f : fminf(far, f2);
    }
}

// MODE 0: any grid (march_next), 1: one cascade, power-of-two grid, linear bitfield copy, 2: the same with dt_gamma == 0
template <int MODE>
__device__ __forceinline__ bool geo_march(const GeoRayArgs& a, const RayGeom& r, float far, float& t, float& x, float& y, float& z, float& dt,
                                          float* t_at = nullptr) {
    if constexpr (MODE == 0) return march_next(a.mk, r, far, t, x, y, z, dt, t_at);
    else return march_next_c1<MODE == 2>(a.mk, a.linear_grid, a.log2H, r, far, t, x, y, z, dt, t_at);
}

// one visit of the marcher's loop (march_core.hip.h: march_visit*): the flat loops below advance every lane by one visit per
// iteration instead of letting the lanes of a wave wait for each other inside per-sample marching loops
template <int MODE>
__device__ __forceinline__ bool geo_visit(const GeoRayArgs& a, const RayGeom& r, float& t, float& x, float& y, float& z, float& dt) {
    if constexpr (MODE == 0) return a.mk.H_pow2 ? march_visit<true>(a.mk, r, t, x, y, z, dt) : march_visit<false>(a.mk, r, t, x, y, z, dt);
    else return march_visit_c1<MODE == 2>(a.mk, a.linear_grid, a.log2H, r, t, x, y, z, dt);
}

__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
    return v;
}

// linear_grid bit (x + H y + H^2 z) = Morton-ordered bitfield bit morton(x, y, z); one thread per output byte
__global__ void __launch_bounds__(kBlock) k_linearize_bitfield(const uint8_t* __restrict__ grid, uint8_t* __restrict__ linear_grid, uint32_t log2H) {
    const uint32_t byte = blockIdx.x * blockDim.x + threadIdx.x;
    if (byte >= (1u << (3 * log2H)) / 8u) return;
    const uint32_t mask = (1u << log2H) - 1u;
    uint32_t v = 0;
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
        const uint32_t cell = byte * 8u + j;
        const uint32_t m = morton_encode(cell & mask, (cell >> log2H) & mask, cell >> (2 * log2H));
        v |= ((grid[m >> 3] >> (m & 7)) & 1u) << j;
    }
    linear_grid[byte] = (uint8_t)v;
}

// bounding box of the occupied cells -> counters[kCntOcc ..] (all six as maxima, see there).  A few waves, each over a
// contiguous slice of the linear grid and ONE atomic per bound per wave: thousands of waves taking same-address atomics at
// once (the reduction fused into the kernel above) cost 30 us.
constexpr uint32_t kOccBoxWaves = 64;
__global__ void __launch_bounds__(64) k_occupied_box(const uint8_t* __restrict__ linear_grid, uint32_t log2H, uint32_t* __restrict__ counters) {
    const uint32_t bytes = (1u << (3 * log2H)) / 8u, mask = (1u << log2H) - 1u;
    const uint32_t per_wave = ((bytes + kOccBoxWaves - 1) / kOccBoxWaves + 15u) & ~15u;       // whole 16-byte loads (bytes is a multiple of 64)
    const uint32_t lo = min(bytes, blockIdx.x * per_wave), hi = min(bytes, lo + per_wave);
    uint32_t b[6] = {0, 0, 0, 0, 0, 0};
    auto take = [&](uint32_t byte, uint32_t v) {
        if (!v) return;
        // the eight cells of a byte share y and z (H >= 8): x from the lowest / highest set bit
        const uint32_t cell = byte * 8u, x0 = cell & mask, y = (cell >> log2H) & mask, z = cell >> (2 * log2H);
        b[0] = max(b[0], mask - (x0 + (uint32_t)__builtin_ctz(v))); b[1] = max(b[1], mask - y); b[2] = max(b[2], mask - z);
        b[3] = max(b[3], x0 + 31u - (uint32_t)__builtin_clz(v)); b[4] = max(b[4], y); b[5] = max(b[5], z);
    };
    // 16 bytes per lane per load, all of a lane's loads in flight together (the grid is 64-byte granular for H >= 8)
    for (uint32_t base = lo + threadIdx.x * 16u; base < hi; base += 64u * 16u * 4u) {
        uint4 w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t at = base + (uint32_t)k * 64u * 16u;
            w[k] = at + 16u <= hi ? *reinterpret_cast<const uint4*>(linear_grid + at) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t at = base + (uint32_t)k * 64u * 16u;
            const uint32_t words[4] = {w[k].x, w[k].y, w[k].z, w[k].w};
            if (!(words[0] | words[1] | words[2] | words[3])) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) take(at + 4u * q + j, (words[q] >> (8 * j)) & 0xffu);
        }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const uint32_t m = wave_max(b[q]);
        if (threadIdx.x == 0 && m) atomicMax(&counters[kCntOcc + q], m);
    }
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// One round of the per-ray half of the geometry pass; one wave per block.
//   FIRST : slab test, first-hit search (empty-space skipping); rays without a sample are finished here
//   else  : the chunk evaluated by k_geo_eval since the last round is composited (raymarching.cu:996-1030 recurrence,
//           termination tested with the pre-update transmittance), one record per composited sample is appended,
//           finished rays write their outputs
// then every ray still alive marches its next `chunk` samples (march_core.hip.h; the ray time is resumed from the
// composited time after EVERY sample, as the reference loop does with one sample per iteration -- that time does not
// depend on the densities, so marching ahead of the compositor changes nothing) into freshly allocated slots.
constexpr uint32_t kTimeCache = 32;      // the first this many samples of an un-hinted chunk are marched once
constexpr uint32_t kSlotTable = 128;     // samples per ray whose slots the flat write loop can address (per-wave table in LDS)

// Chunk prediction.  A ray that is still alive after compositing n samples has transmittance T >= T_thresh left and saw the
// opacity `alpha` at its last sample.  If the opacity stayed there, the compositor (which stops AFTER the first sample whose
// incoming transmittance is below the threshold) would take j + 1 more samples, j = ceil(ln(T_thresh / T) / ln(1 - alpha)).
// Near a surface of an SDF-derived density the opacity only rises along the ray, so this over-estimates a little and the ray
// is usually finished by the chunk it sizes; in thin media (tiny alpha) the prediction is huge and the cap -- what the ray
// has composited so far, i.e. its sample count at most doubles per round -- decides, which is the schedule a ray without
// any information gets.  Results never depend on the chunking (the marcher is resumed from the composited time after every
// sample whatever the chunks are); what depends on it is how many samples are evaluated past a ray's end: 1.17x the
// composited ones with chunks that grow by half regardless of the ray, 1.06x with this (tests/tools/chunk_sim.py on the
// benchmark scenes), in fewer rounds.
constexpr uint32_t kMinChunk = 2;
__device__ __forceinline__ uint32_t predicted_chunk(float T, float alpha, float T_thresh, uint32_t n_taken, uint32_t most) {
    if (!(T > T_thresh)) return 1u;               // already below the threshold: the compositor takes exactly one more sample
    const uint32_t cap = min(max(8u, n_taken), most);
    if (!(alpha > 1e-6f)) return cap;
    if (alpha >= 1.0f) return kMinChunk;
    const float j = ceilf(__logf(T_thresh / T) / __logf(1.0f - alpha));          // both logarithms negative
    return (uint32_t)fminf(fmaxf(j + 2.0f, (float)kMinChunk), (float)cap);     // j + 1 samples, + 1 of margin
}

template <bool FIRST, int MODE>
__global__ void __launch_bounds__(kBlock) k_geo_rays(const GeoRayArgs a) {
    // ray times of the samples the counting pass found (un-hinted chunks), so that the write pass does not march again
    __shared__ float s_time[kBlock / 64][kTimeCache][64];
    // slot layout of the chunk being written, for the flat write loop: entry i describes sample index c0 + i of the block:
    // the lanes that have such a sample, and how many slots the samples before it take
    __shared__ unsigned long long s_mask[kBlock / 64][kSlotTable];
    __shared__ uint32_t s_before[kBlock / 64][kSlotTable];
    const uint32_t lane = threadIdx.x & 63;
    float* const t_cache = &s_time[threadIdx.x >> 6][0][lane];
    unsigned long long* const tab_mask = s_mask[threadIdx.x >> 6];
    uint32_t* const tab_before = s_before[threadIdx.x >> 6];
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t n_in = FIRST ? (a.N + 63u) / 64u : __builtin_amdgcn_readfirstlane(a.counters[kCntAlive + a.round]);
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    if (((blockIdx.x * blockDim.x + threadIdx.x) >> 6) >= n_in) return;
    const Aabb occ = occupied_box(a);
    for (uint32_t bi = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; bi < n_in; bi += n_waves) {
        const uint32_t blk = FIRST ? bi : __builtin_amdgcn_readfirstlane(a.alive_in[bi]);
        // a ray's id is its position in the caller's list (every input and output is indexed by it); the per-ray state of the
        // pass lives at the block's own 64 slots
        uint32_t ray = blk * 64u + lane;
        if (a.tile_w) ray = ((blk / a.tiles_x) * 8u + (lane >> 3)) * a.tile_w + (blk % a.tiles_x) * 8u + (lane & 7u);
        const bool in_range = ray < a.N;
        uint32_t* const sp = a.state + (blk * 64u + lane);
        RayGeom rg = {};
        float near = 0, far = 0, t_first = 0;
        RayState st = {};
        uint32_t chunk_next = a.chunk;   // samples this ray marches next
        bool alive = false;          // still needs samples after this round's compositing
        bool finish = false;         // write the ray's outputs now
        if constexpr (FIRST) {
            if (in_range) {
                rg = load_ray(a.rays_o, a.rays_d, ray);
                ray_range(a, occ, rg, near, far);
                float t = near, x, y, z, dt;
                const bool hit = (!a.ray_mask || a.ray_mask[ray]) && geo_march<MODE>(a, rg, far, t, x, y, z, dt, &t_first);
                st.acc_t = near;
                alive = hit;
                finish = !hit;
            }
        } else {
            // ---- composite the previous chunk ------------------------------------------------------------------
            // everything a lane needs is loaded in one batch (counts, the state fields, the ray): the fields of a finished ray
            // are stale and unused, but waiting for the counts before asking for the rest doubled the latency of this prologue
            uint32_t cnt = 0, alloc = 0;
            bool last = false;
            if (in_range) {
                const uint32_t cc = sp[kFChunk * (size_t)a.n_pad];
                alloc = sp[kFAlloc * (size_t)a.n_pad];
                uint32_t f[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) f[q] = sp[(size_t)q * a.n_pad];          // kFAccT .. kFTaken
                static_assert(kFAccT == 0 && kFTaken == 7 && kFChunk == 8, "state field order");
                rg = load_ray(a.rays_o, a.rays_d, ray);
                cnt = cc & 0x7fffffffu;
                last = (cc >> 31) != 0;
                st.acc_t = __uint_as_float(f[kFAccT]); st.ws = __uint_as_float(f[kFWs]); st.depth = __uint_as_float(f[kFDepth]);
                st.an[0] = __uint_as_float(f[kFAn0]); st.an[1] = __uint_as_float(f[kFAn1]); st.an[2] = __uint_as_float(f[kFAn2]);
                st.arough = __uint_as_float(f[kFRough]); st.n_taken = f[kFTaken];
            }
            const bool active = cnt != 0;
            if (active) ray_range(a, occ, rg, near, far);
            else st = RayState{};
            const uint32_t base = __builtin_amdgcn_readfirstlane(a.block_base[blk]);
            uint32_t k = 0;              // samples of the chunk that get composited
            bool terminated = false;
            float last_alpha = 0.0f;
            // Both passes below walk the chunk in groups of kGroup samples: the slot addresses of a group depend only on the
            // per-ray counts, so its loads are issued together and ONE memory latency is paid per group instead of one per
            // sample (the arrays were written by k_geo_eval32 on other XCDs: every load comes from the fabric, ~2 us under
            // load, and a wave's chain of them WAS this kernel's run time).  Lanes without a sample read the block's first
            // slot (a valid address) and ignore the value; the arithmetic and its order are unchanged.
            constexpr uint32_t kGroup = 8;
            {
                float ws = st.ws;        // pass 1: how many records does each ray append
                bool going = active;
                uint32_t off = 0;
                for (uint32_t c0 = 0;; c0 += kGroup) {
                    if (!__ballot(going && c0 < cnt)) break;
                    float al[kGroup];
#pragma unroll
                    for (uint32_t j = 0; j < kGroup; ++j) {
                        const uint32_t c = c0 + j;
                        const unsigned long long m = __ballot(c < alloc);
                        const uint32_t slot = base + off + (uint32_t)__popcll(m & below);
                        off += (uint32_t)__popcll(m);
                        al[j] = a.alpha[c < cnt ? slot : base];
                    }
#pragma unroll
                    for (uint32_t j = 0; j < kGroup; ++j) {
                        if (going && c0 + j < cnt) {
                            const float T = 1 - ws;
                            ws += al[j] * T;
                            ++k;
                            if (T < a.T_thresh) { terminated = true; going = false; }
                        }
                    }
                }
            }
            const uint32_t wave_records = wave_sum(k);
            uint32_t rec_base = 0;
            if (lane == 0 && wave_records) rec_base = atomicAdd(&a.counters[kCntRecordHead], wave_records);
            rec_base = __builtin_amdgcn_readfirstlane(rec_base);
            const bool fits = rec_base + wave_records <= a.rec_cap;
            if (!fits && lane == 0) a.counters[kCntOverflow] = 1u;
            {
                typedef float f32x3 __attribute__((ext_vector_type(3), aligned(4)));   // 12-byte stride arrays: NOT 16-byte aligned
                constexpr uint32_t kGroup2 = 4;
                uint32_t off = 0, roff = 0;
                for (uint32_t c0 = 0;; c0 += kGroup2) {            // pass 2: the recurrence itself + the records
                    if (!__ballot(c0 < k)) break;
                    uint32_t slot[kGroup2], rec[kGroup2];
                    float al[kGroup2], dd[kGroup2], ro[kGroup2];
                    f32x3 nv[kGroup2];
#pragma unroll
                    for (uint32_t j = 0; j < kGroup2; ++j) {
                        const uint32_t c = c0 + j;
                        const unsigned long long m = __ballot(c < alloc), mk = __ballot(c < k);
                        slot[j] = base + off + (uint32_t)__popcll(m & below);
                        rec[j] = rec_base + roff + (uint32_t)__popcll(mk & below);
                        off += (uint32_t)__popcll(m);
                        roff += (uint32_t)__popcll(mk);
                        const uint32_t at = c < k ? slot[j] : base;
                        al[j] = a.alpha[at];
                        dd[j] = a.dd[at];
                        ro[j] = a.rough[at];
                        nv[j] = *reinterpret_cast<const f32x3*>(a.normal + 3 * (size_t)at);
                    }
#pragma unroll
                    for (uint32_t j = 0; j < kGroup2; ++j) {
                        if (c0 + j < k) {
                            const float alpha = al[j];
                            last_alpha = alpha;
                            const float T = 1 - st.ws;
                            const float w = alpha * T;
                            st.ws += w;
                            st.acc_t = st.acc_t + dd[j];
                            st.depth += w * st.acc_t;
#pragma unroll
                            for (int d = 0; d < 3; ++d) st.an[d] += w * nv[j][d];
                            st.arough += w * ro[j];
                            const uint32_t r = rec[j];
                            if (fits) { a.rec_ray[r] = ray; a.rec_idx[r] = st.n_taken; a.rec_w[r] = w; a.rec_slot[r] = slot[j]; }
                            ++st.n_taken;
                        }
                    }
                }
            }
            if (active) {
                finish = terminated || last || st.n_taken >= a.max_samples;
                alive = !finish;
            }
            if (alive && a.predict) chunk_next = predicted_chunk(1 - st.ws, last_alpha, a.T_thresh, st.n_taken, a.chunk);
        }

        // ---- march the next chunk ------------------------------------------------------------------------------
        // Round 0 sizes a ray's first chunk from the caller's hint when there is one (samples the ray took in an earlier
        // render of about the same camera: exact for the frames of a fixed-camera video): the ray is then done in one
        // round with nothing evaluated past its end.  A wrong hint costs a round or some wasted samples, never accuracy.
        uint32_t chunk = chunk_next;
        bool lane_hinted = false;
        if constexpr (FIRST) {
            if (alive && a.ray_cost) { const uint32_t h = a.ray_cost[ray]; if (h) { chunk = min(h, 4096u); lane_hinted = true; } }
        }
        // With a hint the ray's slots are sized by it and the ray is marched once (slots the ray turns out not to need
        // are zero-filled: evaluated, never composited); without one the marcher first counts.  The choice is made per WAVE
        // (the sample-major slot layout is walked by the whole wave together): a block in which some live ray has no hint --
        // a silhouette of a moving camera -- counts, with the hints of the others as their chunk sizes.
        const bool hinted = FIRST && !__any(alive && !lane_hinted);
        // Marching is written as FLAT loops: every iteration each lane that still has something to do makes ONE visit of the
        // marcher's loop (an occupied cell = a sample; an empty cell = a hop to its exit).  With a marching loop per sample
        // inside a loop over samples, a wave pays, for every sample index, the longest walk any of its lanes makes there --
        // one ray leaving the shell and crossing the hollow interior stalls the other 63 for ~60 visits, and in a block on the
        // silhouette a different lane does so at almost every index: ~1000 visits per block instead of ~190, and the kernel
        // ran as long as its worst block (tools/sim/march_sim.py).
        uint32_t want = 0;           // slots to allocate
        if (alive && a.chunk) {
            if (hinted) want = min(chunk, a.max_samples - min(st.n_taken, a.max_samples));
        }
        if (!hinted) {
            // counting pass: how many samples can the ray take (at most `chunk`); the ray times of the first kTimeCache of
            // them stay in LDS, the write pass below does not march again for those
            const uint32_t most = (alive && a.chunk) ? min(chunk, a.max_samples - min(st.n_taken, a.max_samples)) : 0u;
            float tr = st.acc_t, t = FIRST ? t_first : st.acc_t;
            bool going = most != 0;
            while (__any(going)) {
                if (going) {
                    if (!(t < far)) going = false;
                    else {
                        float x, y, z, dt;
                        if (geo_visit<MODE>(a, rg, t, x, y, z, dt)) {
                            if (want < kTimeCache) t_cache[want * 64] = t;
                            const float ta = t + dt;
                            tr = tr + (ta - tr);
                            t = tr;                              // the reference resumes from the composited ray time
                            going = ++want < most;
                        }
                    }
                }
            }
        }
        const uint32_t wave_slots = wave_sum(want);
        uint32_t base = 0;
        if (lane == 0 && wave_slots) base = atomicAdd(&a.counters[kCntSampleHead], wave_slots);
        base = __builtin_amdgcn_readfirstlane(base);
        if (wave_slots && base + wave_slots > a.cap) {
            if (lane == 0) a.counters[kCntOverflow] = 2u;
            want = 0;
        }
        uint32_t marched = 0;
        {
            float tr = st.acc_t, t_head = FIRST ? t_first : st.acc_t;
            bool ended = false;
            uint32_t off = 0, c = 0;
            auto store = [&](size_t slot, float x, float y, float z, float dt, float dd) {
                a.xyz[3 * slot] = x; a.xyz[3 * slot + 1] = y; a.xyz[3 * slot + 2] = z;
                a.dt[slot] = dt;
                a.dd[slot] = dd;
            };
            // (A) samples whose ray times the counting pass left in LDS: no marching, the wave walks the sample indices together
            const uint32_t n_cached = hinted ? 0u : kTimeCache;
            for (; c < n_cached; ++c) {
                const unsigned long long m = __ballot(c < want);
                if (!m) break;
                const size_t slot = (size_t)base + off + (uint32_t)__popcll(m & below);
                off += (uint32_t)__popcll(m);
                if (c < want) {
                    // position, step and ray-time bookkeeping are the marcher's own statements for an occupied cell
                    const float t_at = t_cache[c * 64];
                    const float x = clampf(rg.ox + t_at * rg.dx, -a.mk.bound, a.mk.bound);
                    const float y = clampf(rg.oy + t_at * rg.dy, -a.mk.bound, a.mk.bound);
                    const float z = clampf(rg.oz + t_at * rg.dz, -a.mk.bound, a.mk.bound);
                    const float dt = step_size(a.mk, t_at);
                    const float t = t_at + dt;
                    const float dd = t - tr;
                    tr = tr + dd;
                    t_head = tr;
                    ++marched;
                    store(slot, x, y, z, dt, dd);
                }
            }
            // (B) the rest is marched here.  The slot of (lane, sample index) depends on the other lanes' counts: a per-wave
            // table of the next kSlotTable sample indices -- which lanes have that sample, how many slots come before it --
            // lets every lane advance at its own pace, one visit per iteration; rays that end before their hint said
            // (an over-estimating hint) zero-fill what is left of their slots.
            while (true) {
                const uint32_t c0 = c;                                        // wave-uniform: first sample index of this table
                const unsigned long long some = __ballot(c0 < want);
                if (!some) break;
                uint32_t filled = 0;
                for (uint32_t i = 0; i < kSlotTable; ++i) {
                    const unsigned long long m = __ballot(c0 + i < want);
                    if (!m) break;
                    if (lane == 0) { tab_mask[i] = m; tab_before[i] = off; }
                    off += (uint32_t)__popcll(m);
                    filled = i + 1;
                }
                wave_lds_sync();
                const uint32_t c_end = c0 + filled;
                uint32_t ci = c0;                                              // this lane's next sample index
                bool busy = ci < want;
                while (__any(busy)) {
                    if (busy) {
                        float x = 0, y = 0, z = 0, dt = 0, dd = 0;
                        bool emit = ended;
                        if (!ended) {
                            if (!(t_head < far)) { ended = true; emit = true; }
                            else if (geo_visit<MODE>(a, rg, t_head, x, y, z, dt)) {
                                const float ta = t_head + dt;
                                dd = ta - tr;
                                tr = tr + dd;
                                t_head = tr;
                                ++marched;
                                emit = true;
                            }
                        }
                        if (emit) {
                            if (ended) x = y = z = dt = dd = 0;
                            const uint32_t i = ci - c0;
                            store((size_t)base + tab_before[i] + (uint32_t)__popcll(tab_mask[i] & below), x, y, z, dt, dd);
                            ++ci;
                            busy = ci < want && ci < c_end;
                        }
                    }
                }
                wave_lds_sync();
                c = c_end;
            }
        }
        if (alive) {
            st.chunk_count = marched | ((marched < chunk) ? 0x80000000u : 0u);
            if (marched == 0) { finish = true; alive = false; }   // nothing left along the ray (or the last round: cannot
        }                                                         // leave anything, the chunk schedule covers max_steps)
        // next round's block list + the per-ray state of the blocks on it
        {
            const unsigned long long m = __ballot(alive);
            if (m) {
                if (lane == 0) {
                    a.alive_out[atomicAdd(&a.counters[kCntAlive + a.round + 1], 1u)] = blk;
                    a.block_base[blk] = base;
                }
                if (in_range) {
                    sp[kFChunk * (size_t)a.n_pad] = alive ? st.chunk_count : 0u;
                    sp[kFAlloc * (size_t)a.n_pad] = want;            // every lane's share of the block's slot layout, alive or not
                }
                if (alive) {
                    sp[kFAccT * (size_t)a.n_pad] = __float_as_uint(st.acc_t);
                    sp[kFWs * (size_t)a.n_pad] = __float_as_uint(st.ws);
                    sp[kFDepth * (size_t)a.n_pad] = __float_as_uint(st.depth);
                    sp[kFAn0 * (size_t)a.n_pad] = __float_as_uint(st.an[0]);
                    sp[kFAn1 * (size_t)a.n_pad] = __float_as_uint(st.an[1]);
                    sp[kFAn2 * (size_t)a.n_pad] = __float_as_uint(st.an[2]);
                    sp[kFRough * (size_t)a.n_pad] = __float_as_uint(st.arough);
                    sp[kFTaken * (size_t)a.n_pad] = st.n_taken;
                }
            }
        }
        if (finish) {
            // run_cuda epilogue for this ray (cuda_ray.py:348-362); the colour images are composited later from the records
            const size_t id = ray;
            a.depth[id] = st.depth;
            a.ws[id] = st.ws;
            if (a.nimg) {
                const float inv = 1.0f / fmaxf(sqrtf(st.an[0] * st.an[0] + st.an[1] * st.an[1] + st.an[2] * st.an[2]), 1e-10f);
                a.nimg[3 * id] = st.an[0] * inv; a.nimg[3 * id + 1] = st.an[1] * inv; a.nimg[3 * id + 2] = st.an[2] * inv;
            }
            if (a.rimg) a.rimg[id] = st.arough;
            if (a.ray_cost) a.ray_cost[id] = (uint16_t)min(st.n_taken, 65535u);
        }
    }
}

int fill_eval_args(const envidr_render_desc* d, GeoEvalArgs& a, const char* who) {
    ENVIDR_REQUIRE(d->hash_table && d->sdf_blob && d->sdf_w3_row0, "%s: null hash table / SDF weights", who);
    ENVIDR_REQUIRE(d->num_levels == ENVIDR_MAX_LEVELS, "%s: built for 16 hash levels", who);
    ENVIDR_REQUIRE(d->beta > 0, "%s: beta must be positive", who);
    a.table = d->hash_table;
    a.table_bytes = (uint32_t)d->hash_offsets[d->num_levels] * 8u;
    const char* err = fill_lean_levels(d, a.lv);
    ENVIDR_REQUIRE(!err, "%s: %s", who, err);
    a.bound = d->bound; a.bound2 = 2 * d->bound;
    a.sdf_blob = d->sdf_blob; a.sdf_w3r0 = d->sdf_w3_row0;
    a.sdf_e16_blob = d->sdf_geo_blob;
    a.beta = d->beta; a.inv_beta = 1 / d->beta; a.density_scale = d->density_scale;
    a.rough_bias = d->roughness_bias; a.rough_act_scale = d->roughness_act_scale; a.rough_scale = d->roughness_scale;
    return ENVIDR_OK;
}

void launch_eval(const GeoEvalArgs& a, uint32_t max_samples, hipStream_t s) {
    // one workgroup per CU (the LDS-resident weights and the parked Jacobians fill the CU's LDS), persistent over the batches
    if (a.sdf_e16_blob) {
        const uint32_t blocks16 = std::max(1u, std::min((uint32_t)device_cu_count(), ceil_div(max_samples, kE16Waves * 16u)));
        hipLaunchKernelGGL(k_geo_eval16, dim3(blocks16), dim3(kE16Threads), 0, s, a);
        return;
    }
    const uint32_t blocks32 = std::max(1u, std::min((uint32_t)device_cu_count(), ceil_div(max_samples, kE32Waves * 32u)));
    if (a.probe_feat || a.probe_rows || a.probe_raw || a.probe_grad) hipLaunchKernelGGL(k_geo_eval32<true>, dim3(blocks32), dim3(kE32Threads), 0, s, a);
    else hipLaunchKernelGGL(k_geo_eval32<false>, dim3(blocks32), dim3(kE32Threads), 0, s, a);
}


uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

struct GeoLayout {
    uint64_t counters, alive0, alive1, block_base, state, linear_grid, xyz, dt, dd, alpha, normal, geo, rough, blend, total;
    uint32_t n_pad;
};
constexpr uint64_t kLinearGridBytes = 256ull * 256 * 256 / 8;        // the largest grid the fast marcher takes
GeoLayout geo_layout(uint32_t N, uint32_t cap) {
    GeoLayout L;
    uint64_t o = 0;
    auto take = [&](uint64_t bytes) { const uint64_t at = o; o = align_up(o + bytes, 256); return at; };
    L.counters = take(kGeoCounterWords * 4);
    const uint64_t blocks = ((uint64_t)N + 63) / 64;
    L.n_pad = (uint32_t)(blocks * 64);
    L.alive0 = take(blocks * 4); L.alive1 = take(blocks * 4); L.block_base = take(blocks * 4);
    L.state = take((uint64_t)kStateFields * L.n_pad * 4);
    L.linear_grid = take(kLinearGridBytes);
    L.xyz = take((uint64_t)cap * 12); L.dt = take((uint64_t)cap * 4); L.dd = take((uint64_t)cap * 4);
    L.alpha = take((uint64_t)cap * 4); L.normal = take((uint64_t)cap * 12); L.geo = take((uint64_t)cap * 48);
    L.rough = take((uint64_t)cap * 4); L.blend = take((uint64_t)cap * 4);
    L.total = o;
    return L;
}

__global__ void k_geo_finalize(const uint32_t* __restrict__ counters, uint32_t* __restrict__ rec_counter, unsigned long long* __restrict__ stats) {
    // a frame that did not fit is reported through a record count no capacity can hold: the shading / compositing entry
    // points then do nothing and the host, when it eventually looks, redoes the frame with larger buffers
    *rec_counter = counters[kCntOverflow] ? 0xffffffffu : counters[kCntRecordHead];
    if (stats) { stats[0] = counters[kCntSampleHead]; stats[1] = counters[kCntRecordHead]; stats[2] = counters[kCntOverflow]; }
}

}  // namespace

extern "C" {

uint64_t envidr_geometry_workspace_bytes(uint32_t N, uint32_t sample_capacity) { return geo_layout(N, sample_capacity).total; }

int envidr_geometry_pass(const envidr_render_desc* d, const float* rays_o, const float* rays_d, uint32_t N,
                         const envidr_render_out* out, envidr_geometry_export* rec, void* workspace, uint64_t workspace_bytes,
                         uint32_t sample_capacity, envidr_stream_t stream) {
    ENVIDR_REQUIRE(d && out && rec, "geometry_pass: null descriptor");
    if (N == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(rays_o && rays_d && workspace, "geometry_pass: null ray pointers / workspace");
    ENVIDR_REQUIRE(out->depth && out->weights_sum, "geometry_pass: depth and weights_sum are required");
    ENVIDR_REQUIRE(d->density_bitfield, "geometry_pass: null bitfield");
    ENVIDR_REQUIRE(d->cascades >= 1 && d->grid_size >= 1 && d->max_steps >= 1, "geometry_pass: bad grid parameters");
    ENVIDR_REQUIRE(rec->counter && rec->ray && rec->idx && rec->w && rec->slot && rec->capacity, "geometry_pass: null pointer in the record arrays");
    ENVIDR_REQUIRE(sample_capacity >= 64, "geometry_pass: sample_capacity too small");
    const GeoLayout L = geo_layout(N, sample_capacity);
    ENVIDR_REQUIRE(workspace_bytes >= L.total, "geometry_pass: workspace of %llu bytes, need %llu", (unsigned long long)workspace_bytes,
                   (unsigned long long)L.total);
    ENVIDR_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "geometry_pass: workspace must be 256-byte aligned");
    char* ws = reinterpret_cast<char*>(workspace);
    hipStream_t s = as_stream(stream);

    GeoEvalArgs e;
    memset(&e, 0, sizeof(e));
    {
        const int rc = fill_eval_args(d, e, "geometry_pass");
        if (rc) return rc;
    }
    uint32_t* counters = reinterpret_cast<uint32_t*>(ws + L.counters);
    e.xyz = reinterpret_cast<float*>(ws + L.xyz); e.dt = reinterpret_cast<float*>(ws + L.dt);
    e.M = sample_capacity; e.head = counters + kCntSampleHead;
    e.alpha = reinterpret_cast<float*>(ws + L.alpha); e.normal = reinterpret_cast<float*>(ws + L.normal);
    e.geo = reinterpret_cast<float*>(ws + L.geo); e.rough = reinterpret_cast<float*>(ws + L.rough); e.blend = reinterpret_cast<float*>(ws + L.blend);

    GeoRayArgs a;
    memset(&a, 0, sizeof(a));
    a.rays_o = rays_o; a.rays_d = rays_d; a.N = N;
    a.mk = make_march_consts(d->bound, d->dt_gamma, d->max_steps, d->cascades, d->grid_size, d->density_bitfield);
    a.box = make_aabb(d); a.min_near = d->min_near; a.T_thresh = d->T_thresh; a.max_samples = d->max_steps;
    a.counters = counters;
    a.state = reinterpret_cast<uint32_t*>(ws + L.state); a.n_pad = L.n_pad;
    a.block_base = reinterpret_cast<uint32_t*>(ws + L.block_base);
    a.xyz = const_cast<float*>(e.xyz); a.dt = const_cast<float*>(e.dt); a.dd = reinterpret_cast<float*>(ws + L.dd);
    a.cap = sample_capacity;
    a.alpha = e.alpha; a.normal = e.normal; a.rough = e.rough;
    a.rec_ray = rec->ray; a.rec_idx = rec->idx; a.rec_w = rec->w; a.rec_slot = const_cast<uint32_t*>(rec->slot); a.rec_cap = rec->capacity;
    a.depth = out->depth; a.ws = out->weights_sum; a.nimg = out->normal_image; a.rimg = out->roughness_image;
    a.ray_cost = d->ray_cost;
    a.ray_mask = d->ray_mask;
    if (d->image_width && d->image_width % 8u == 0 && N % d->image_width == 0 && (N / d->image_width) % 8u == 0) {
        a.tile_w = d->image_width;
        a.tiles_x = d->image_width / 8u;
    }
    uint32_t* alive[2] = {reinterpret_cast<uint32_t*>(ws + L.alive0), reinterpret_cast<uint32_t*>(ws + L.alive1)};

    // Chunk schedule.  Round 0: the caller's per-ray hint, else 16 samples.  Rounds 1..5: each live ray sizes its own chunk
    // (predicted_chunk: at least kMinChunk, at most what it has composited so far).  A ray still alive after six rounds is
    // crossing something thin or translucent and is given ALL that is left in one go (the counting pass allocates only what
    // it can actually march, bounded by max_steps - composited): whatever the hints and predictions were, seven rounds
    // cover max_steps, and a frame whose rays are all done after the first round or two -- the hinted frames of a video --
    // pays for few empty launches.
    constexpr uint32_t kRounds = 7;
    static_assert(kRounds + 2 <= kMaxRounds, "round-indexed counters");
    uint32_t chunks[kRounds];
    for (uint32_t r = 0; r < kRounds; ++r) chunks[r] = r == 0 ? std::min(16u, d->max_steps) : d->max_steps;
    const uint32_t rounds = kRounds;

    if (hipMemsetAsync(counters, 0, kGeoCounterWords * 4, s) != hipSuccess) return check_launch("geometry_pass memset");
    // one cascade on a power-of-two grid (every scene of the reference): the marcher reads a linear-order copy of the bitfield
    int mode = 0;
    if (d->cascades == 1 && (d->grid_size & (d->grid_size - 1)) == 0 && d->grid_size >= 8 && d->grid_size <= 256) {
        uint32_t log2H = 0;
        while ((1u << log2H) < d->grid_size) ++log2H;
        uint8_t* lin = reinterpret_cast<uint8_t*>(ws + L.linear_grid);
        const uint32_t bytes = (1u << (3 * log2H)) / 8u;
        hipLaunchKernelGGL(k_linearize_bitfield, dim3(ceil_div(bytes, kBlock)), dim3(kBlock), 0, s, d->density_bitfield, lin, log2H);
        a.linear_grid = lin; a.log2H = log2H;
        a.occ_clip = 1;
        for (int i = 0; i < 3; ++i)
            if (a.box.lo[i] < -d->bound || a.box.hi[i] > d->bound) a.occ_clip = 0;       // positions outside the cube are clamped INTO boundary cells
        if (a.occ_clip) hipLaunchKernelGGL(k_occupied_box, dim3(kOccBoxWaves), dim3(64), 0, s, lin, log2H, counters);
        mode = d->dt_gamma == 0 ? 2 : 1;
    }
    const uint32_t ray_blocks = ceil_div(N, kBlock);
    for (uint32_t r = 0; r <= rounds; ++r) {
        a.round = r;
        a.chunk = r < rounds ? chunks[r] : 0u;
        a.predict = (r >= 1 && r + 1 < rounds) ? 1u : 0u;
        a.alive_in = alive[(r + 1) & 1];
        a.alive_out = alive[r & 1];
        const dim3 grid(r == 0 ? ray_blocks : std::min(ray_blocks, 1024u));
        if (r == 0) {
            if (mode == 2) hipLaunchKernelGGL((k_geo_rays<true, 2>), grid, dim3(kBlock), 0, s, a);
            else if (mode == 1) hipLaunchKernelGGL((k_geo_rays<true, 1>), grid, dim3(kBlock), 0, s, a);
            else hipLaunchKernelGGL((k_geo_rays<true, 0>), grid, dim3(kBlock), 0, s, a);
        } else {
            if (mode == 2) hipLaunchKernelGGL((k_geo_rays<false, 2>), grid, dim3(kBlock), 0, s, a);
            else if (mode == 1) hipLaunchKernelGGL((k_geo_rays<false, 1>), grid, dim3(kBlock), 0, s, a);
            else hipLaunchKernelGGL((k_geo_rays<false, 0>), grid, dim3(kBlock), 0, s, a);
        }
        {
            const int rc = check_launch("k_geo_rays");
            if (rc) return rc;
        }
        if (r < rounds) {
            e.begin_io = counters + kCntBegin + r;
            launch_eval(e, sample_capacity, s);
            const int rc = check_launch("k_geo_eval");
            if (rc) return rc;
        }
    }
    hipLaunchKernelGGL(k_geo_finalize, dim3(1), dim3(1), 0, s, counters, rec->counter, reinterpret_cast<unsigned long long*>(out->stats));
    // where the records' per-sample data lives (indexed through rec->slot)
    rec->normal = e.normal; rec->geo_feat = e.geo; rec->roughness = e.rough; rec->blend = e.blend;
    rec->image_width = a.tile_w;
    return check_launch("k_geo_finalize");
}


uint32_t envidr_sdf_geometry_floats(void) { return (uint32_t)(kE16BlobFloats + kE16W3Row); }

int envidr_pack_sdf_geometry(const float* W1, const float* b1, const float* W2, const float* b2, const float* W3, const float* b3, float* dst_host) {
    ENVIDR_REQUIRE(W1 && b1 && W2 && b2 && W3 && b3 && dst_host, "pack_sdf_geometry: null pointer");
    pack_sdf_e16(W1, b1, W2, b2, W3, b3, dst_host);
    return ENVIDR_OK;
}

int envidr_geometry_eval(const envidr_render_desc* d, const float* xyz, const float* dt, uint32_t M, const uint32_t* range_dev,
                         const envidr_geometry_samples_out* out, envidr_stream_t stream) {
    ENVIDR_REQUIRE(d && out, "geometry_eval: null descriptor");
    if (M == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(xyz && (dt || !out->alpha), "geometry_eval: null sample pointers");
    GeoEvalArgs a;
    memset(&a, 0, sizeof(a));
    const int rc = fill_eval_args(d, a, "geometry_eval");
    if (rc) return rc;
    a.xyz = xyz; a.dt = dt; a.M = M; a.range = range_dev;
    a.alpha = out->alpha; a.sigma = out->sigma; a.normal = out->normal; a.geo = out->geo_feat; a.rough = out->roughness; a.blend = out->blend;
    launch_eval(a, M, as_stream(stream));
    return check_launch("k_geo_eval");
}


int envidr_geometry_probe(const envidr_render_desc* d, const float* xyz, uint32_t M, float* features, uint32_t* corner_rows,
                          float* raw_outputs, float* sdf_gradient, envidr_stream_t stream) {
    ENVIDR_REQUIRE(d, "geometry_probe: null descriptor");
    if (M == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(xyz && (features || corner_rows || raw_outputs || sdf_gradient), "geometry_probe: null sample pointer / nothing to export");
    GeoEvalArgs a;
    memset(&a, 0, sizeof(a));
    const int rc = fill_eval_args(d, a, "geometry_probe");
    if (rc) return rc;
    a.sdf_e16_blob = nullptr;                   // the probe instantiation exists for k_geo_eval32, the kernel the frames run
    a.xyz = xyz; a.M = M;
    a.probe_feat = features; a.probe_rows = corner_rows; a.probe_raw = raw_outputs; a.probe_grad = sdf_gradient;
    launch_eval(a, M, as_stream(stream));
    return check_launch("k_geo_eval32<probe>");
}

}  // extern "C"
