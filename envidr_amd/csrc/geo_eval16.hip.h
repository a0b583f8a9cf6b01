// k_geo_eval16: the per-sample geometry evaluation (geometry_pass.hip) on 16-column MFMAs: 16 samples per wave, the 16 hash
// levels split between the four lane QUARTERS (lane = 16 q + n evaluates levels 4 i + q of sample n).
//
// Why: the evaluation has two costs of the same size -- the hash sweep (bound by the rate at which L2 misses come back:
// ~1.95 ms per 7.7 M samples) and the SDF network forward + backward on the matrix cores (~1.5 ms) -- and they only overlap
// across waves.  With 32 x 32 x 2 MFMAs a wave holds 32 samples, its parked Jacobian is 12 KiB, and two waves per SIMD is all
// that fits beside the LDS-resident weights (k_geo_eval32: 2.75 ms).  `v_mfma_f32_16x16x4_f32` has the same rate (2048 FLOP
// in 8 passes), a quarter of the accumulator registers per tile and 16 columns: a wave's Jacobian is 6 KiB, its register
// need drops below a third of the file, and three to four waves per SIMD take turns on the matrix pipe while the others gather.
//
// Operand layout of v_mfma_f32_16x16x4_f32: A[m][k] in lane (m = lane & 15, k = lane >> 4), B[k][n] in lane (n = lane & 15,
// k = lane >> 4), D[4 q + r][n] in register r of lane (n, q).  As with the 32-column scheme the weights are the A operand and
// the reduction order of every layer is chosen so that the B operand of a step is a register the lane already holds:
//   layer 1   step (i, c), slot q  =  feature 2 (4 i + q) + c   -- the lane's own level 4 i + q, channel c
//   layers fed by accumulator tiles: step (T, r), slot q  =  feature 16 T + 4 q + r
// and the ROWS of the two layers whose outputs go back to per-lane work are permuted on the host so that they land where
// they are needed without any cross-lane traffic:
//   W1^T (feature gradients): tile T, row 4 q + r  =  d sdf / d feature 2 (4 (2 T + (r >> 1)) + q) + (r & 1): quarter q gets the
//        gradients of exactly the levels whose Jacobian it parked
//   W3 (15 outputs): quarter 0 rows {sdf, geo 0..2}, quarter 1 {roughness, geo 3..5}, quarter 2 {blend logit, geo 6..8},
//        quarter 3 {-, geo 9..11}: every quarter stores its own three features and one scalar
// The only cross-lane steps are two butterfly adds (lanes ^ 16, ^ 32) for the normal and for |geo|^2.
#pragma once

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// fragment counts ([step][tile], bias = one extra leading step)
constexpr int kE16L1 = 0, kE16L2 = kE16L1 + (8 + 1) * 4, kE16L3 = kE16L2 + (16 + 1) * 4, kE16B2 = kE16L3 + (16 + 1) * 1,
              kE16B1 = kE16B2 + 16 * 4, kE16Frags = kE16B1 + 16 * 2;                      // 217
constexpr int kE16Padded = (kE16Frags + kGeoRing - 1) / kGeoRing * kGeoRing;
constexpr int kE16BlobFloats = kE16Padded * 64;
constexpr int kE16W3Row = 64;                                  // W3[0, :] in natural order

// original output of W3 held by row 4 q + r of the (single) output tile; -1: padding
__host__ __device__ constexpr int e16_w3_row(int q, int r) { return r == 0 ? (q == 0 ? 0 : q == 1 ? 13 : q == 2 ? 14 : -1) : 1 + 3 * q + (r - 1); }
// input feature whose gradient is row 4 q + r of tile T of the last backward layer
__host__ __device__ constexpr int e16_grad_feature(int T, int q, int r) { return 2 * (4 * (2 * T + (r >> 1)) + q) + (r & 1); }

// host: the five layers as fragments of 64 floats (lane (m = lane & 15, kq = lane >> 4)), consumption order, followed by
// W3[0, :]; dst holds kE16BlobFloats + kE16W3Row floats
inline void pack_sdf_e16(const float* W1, const float* b1, const float* W2, const float* b2, const float* W3, const float* b3, float* dst) {
    for (int i = 0; i < kE16BlobFloats; ++i) dst[i] = 0.0f;
    for (int k = 0; k < kE16W3Row; ++k) dst[kE16BlobFloats + k] = W3[k];             // row 0 of W3 (d sdf / d h2), natural order
    auto put = [&](int frag, int lane, float v) { dst[(size_t)frag * 64 + lane] = v; };
    for (int lane = 0; lane < 64; ++lane) {
        const int m = lane & 15, kq = lane >> 4;
        // layer 1: 32 -> 64 (4 tiles); bias step first
        for (int T = 0; T < 4; ++T) {
            put(kE16L1 + T, lane, kq == 0 ? b1[16 * T + m] : 0.0f);
            for (int s = 0; s < 8; ++s) put(kE16L1 + (1 + s) * 4 + T, lane, W1[(16 * T + m) * 32 + 2 * (4 * (s >> 1) + kq) + (s & 1)]);
        }
        // layer 2: 64 -> 64
        for (int T = 0; T < 4; ++T) {
            put(kE16L2 + T, lane, kq == 0 ? b2[16 * T + m] : 0.0f);
            for (int s = 0; s < 16; ++s) put(kE16L2 + (1 + s) * 4 + T, lane, W2[(16 * T + m) * 64 + 16 * (s >> 2) + 4 * kq + (s & 3)]);
        }
        // layer 3: 64 -> 15, rows permuted
        {
            const int row = e16_w3_row(m >> 2, m & 3);
            put(kE16L3, lane, (kq == 0 && row >= 0) ? b3[row] : 0.0f);
            for (int s = 0; s < 16; ++s) put(kE16L3 + 1 + s, lane, row >= 0 ? W3[row * 64 + 16 * (s >> 2) + 4 * kq + (s & 3)] : 0.0f);
        }
        // backward through layer 2: out[j] = sum_k W2[k][j] g[k]
        for (int T = 0; T < 4; ++T)
            for (int s = 0; s < 16; ++s) put(kE16B2 + s * 4 + T, lane, W2[(16 * (s >> 2) + 4 * kq + (s & 3)) * 64 + 16 * T + m]);
        // backward through layer 1, rows = the gradient of the feature the receiving quarter owns
        for (int T = 0; T < 2; ++T)
            for (int s = 0; s < 16; ++s) put(kE16B1 + s * 2 + T, lane, W1[(16 * (s >> 2) + 4 * kq + (s & 3)) * 32 + e16_grad_feature(T, m >> 2, m & 3)]);
    }
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int NSTEPS, int MT, int F0, int S = 0, class Src, class BOp>
__device__ __forceinline__ void e16_steps(Src& wp, f32x4 (&acc)[MT], BOp&& b) {
    if constexpr (S < NSTEPS) {
        const float bv = b(std::integral_constant<int, S>{});
        [&]<int... T>(std::integer_sequence<int, T...>) {
            ((acc[T] = mfma16(wp.template take<F0 + S * MT + T, kE16Padded>(), bv, acc[T])), ...);
        }(std::make_integer_sequence<int, MT>{});
        __builtin_amdgcn_sched_barrier(0);
        e16_steps<NSTEPS, MT, F0, S + 1>(wp, acc, b);
    }
}
template <int MT>
__device__ __forceinline__ void e16_zero(f32x4 (&acc)[MT]) {
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}
// layer whose input is KT accumulator tiles; RELU applied to the operand cluster of each input tile ahead of its MFMAs
template <int KT, int MT, int F0, bool RELU, bool BIAS, class Src>
__device__ __forceinline__ void e16_layer_from_tiles(Src& wp, uint32_t quarter, const f32x4 (&in)[KT], f32x4 (&acc)[MT]) {
    e16_zero<MT>(acc);
    if constexpr (BIAS) e16_steps<1, MT, F0>(wp, acc, [&](auto) { return quarter == 0 ? 1.0f : 0.0f; });
    [&]<int... K>(std::integer_sequence<int, K...>) {
        ([&] {
            float bq[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) bq[r] = RELU ? relu1(in[K][r]) : in[K][r];
            __builtin_amdgcn_sched_barrier(0);
            e16_steps<4, MT, F0 + (BIAS ? MT : 0) + K * 4 * MT>(wp, acc, [&](auto sc) { return bq[decltype(sc)::value]; });
        }(), ...);
    }(std::make_integer_sequence<int, KT>{});
}

constexpr int kE16Waves = 12;           // per workgroup = per CU: three per SIMD (four if the registers allow)
constexpr int kE16Threads = kE16Waves * 64;
constexpr int kE16LevelSteps = kLevels / 4;

__global__ void __launch_bounds__(kE16Threads, 1) k_geo_eval16(const GeoEvalArgs a) {
    __shared__ __attribute__((aligned(16))) float s_w[kE16BlobFloats + kE16W3Row];
    __shared__ __attribute__((aligned(16))) LeanLevel s_lv[kLevels];
    __shared__ float s_jac[kE16Waves * kE16LevelSteps * 6 * 64];
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t quarter = lane >> 4, sl = lane & 15u;
    uint32_t begin = 0, count = a.M;
    if (a.range) { begin = __builtin_amdgcn_readfirstlane(a.range[0]); count = min(__builtin_amdgcn_readfirstlane(a.range[1]), a.M - min(begin, a.M)); }
    else if (a.head) {
        begin = __builtin_amdgcn_readfirstlane(a.begin_io[0]);
        const uint32_t end = min(__builtin_amdgcn_readfirstlane(*a.head), a.M);
        count = end - min(begin, end);
        if (blockIdx.x == 0 && threadIdx.x == 0) a.begin_io[1] = end;
    }
    const uint32_t batches = (count + 15u) / 16u;
    const uint32_t xcd = blockIdx.x & 7u, in_xcd = blockIdx.x >> 3, per_xcd_blocks = (gridDim.x + 7u - xcd) >> 3;      // XCD-aware batch order, as k_geo_eval32
    const uint32_t share = (batches + 7u) / 8u;
    const uint32_t b_lo = min(xcd * share, batches), b_hi = min(b_lo + share, batches);
    if (b_lo + in_xcd * kE16Waves >= b_hi) return;
    {
        const float4* src = reinterpret_cast<const float4*>(a.sdf_e16_blob);
        float4* dst = reinterpret_cast<float4*>(s_w);
        for (uint32_t i = threadIdx.x; i < (kE16BlobFloats + kE16W3Row) / 4; i += kE16Threads) dst[i] = src[i];
        if (threadIdx.x < kLevels) s_lv[threadIdx.x] = a.lv[threadIdx.x];
    }
    __syncthreads();
    WeightLdsRing<kGeoRing> wp;
    wp.start(s_w + lane);
    const float* w3row = s_w + kE16BlobFloats + 4 * quarter;        // + 16 T: this quarter's four entries of tile T
    float* jac_col = s_jac + wave * (kE16LevelSteps * 6 * 64) + lane;
    constexpr int kAhead = kGeoAhead < kE16LevelSteps ? kGeoAhead : kE16LevelSteps - 1;
    const __amdgpu_buffer_rsrc_t table = table_rsrc(a.table, a.table_bytes);
    const uint32_t stride = per_xcd_blocks * kE16Waves;
    typedef float f32x3 __attribute__((ext_vector_type(3), aligned(4)));   // 12-byte stride arrays: NOT 16-byte aligned

    for (uint32_t b = b_lo + in_xcd * kE16Waves + wave; b < b_hi; b += stride) {
        const uint32_t sidx = b * 16u + sl;
        const bool on = sidx < count;
        const size_t slot = (size_t)begin + (on ? sidx : 0u);
        float xc[3];
        bool inside;
        {
            f32x3 pv = {0.0f, 0.0f, 0.0f};
            if (on) pv = *reinterpret_cast<const f32x3*>(a.xyz + 3 * slot);
            const float x01[3] = {(pv[0] + a.bound) / a.bound2, (pv[1] + a.bound) / a.bound2, (pv[2] + a.bound) / a.bound2};
            inside = x01[0] >= 0 && x01[0] <= 1 && x01[1] >= 0 && x01[1] <= 1 && x01[2] >= 0 && x01[2] <= 1;
#pragma unroll
            for (int d = 0; d < 3; ++d) xc[d] = inside ? x01[d] : 0.5f;
        }

        // ================= phase 1: this lane's four levels: values + Jacobian -> LDS ============================
        float f[kE16LevelSteps][2];
        {
            LeanStage st[kAhead + 1];
            LeanLevel lvs[kAhead + 1];
            auto prep = [&](int i, LeanStage& stg, LeanLevel& lv) {
                lv = s_lv[4 * i + quarter];
                lean_prepare<0, true>(lv, table, xc, stg);
            };
            [&]<int... I>(std::integer_sequence<int, I...>) { (prep(I, st[I], lvs[I]), ...); }(std::make_integer_sequence<int, kAhead>{});
            __builtin_amdgcn_sched_barrier(0);
            auto step = [&](auto ic, LeanStage& now, LeanLevel& lvnow, LeanStage& ahead, LeanLevel& lvahead) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i + kAhead < kE16LevelSteps) prep(i + kAhead, ahead, lvahead);
                __builtin_amdgcn_sched_barrier(0);
                f32x2 o, g[3];
                lean_finish(now, inside ? lvnow.on : 0.0f, o, g);
                f[i][0] = o[0]; f[i][1] = o[1];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    jac_col[((i * 3 + d) * 2 + 0) * 64] = g[d][0];
                    jac_col[((i * 3 + d) * 2 + 1) * 64] = g[d][1];
                }
            };
            [&]<int... I>(std::integer_sequence<int, I...>) {
                (step(std::integral_constant<int, I>{}, st[I % (kAhead + 1)], lvs[I % (kAhead + 1)], st[(I + kAhead) % (kAhead + 1)],
                      lvs[(I + kAhead) % (kAhead + 1)]), ...);
            }(std::make_integer_sequence<int, kE16LevelSteps>{});
        }

        // ================= phase 2: SDF network forward + input gradient for the wave's 16 samples ===================
        f32x4 o3[1], gf[2];
        {
            f32x4 h1[4], h2[4];
            e16_zero<4>(h1);
            e16_steps<1, 4, kE16L1>(wp, h1, [&](auto) { return quarter == 0 ? 1.0f : 0.0f; });
            e16_steps<8, 4, kE16L1 + 4>(wp, h1, [&](auto sc) { constexpr int s = decltype(sc)::value; return f[s >> 1][s & 1]; });
            uint32_t pos1 = 0, pos2 = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) pos1 |= (h1[t][r] > 0 ? 1u : 0u) << (4 * t + r);
            e16_layer_from_tiles<4, 4, kE16L2, true, true>(wp, quarter, h1, h2);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) pos2 |= (h2[t][r] > 0 ? 1u : 0u) << (4 * t + r);
            e16_layer_from_tiles<4, 1, kE16L3, true, true>(wp, quarter, h2, o3);
            f32x4 g2[4], g1[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float4 w = *reinterpret_cast<const float4*>(w3row + 16 * t);
                const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) g2[t][r] = (pos2 >> (4 * t + r)) & 1u ? wv[r] : 0.0f;
            }
            e16_layer_from_tiles<4, 4, kE16B2, false, false>(wp, quarter, g2, g1);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) g1[t][r] = (pos1 >> (4 * t + r)) & 1u ? g1[t][r] : 0.0f;
            e16_layer_from_tiles<4, 2, kE16B1, false, false>(wp, quarter, g1, gf);
            wp.template end_pass<kE16Frags>();
        }

        // ================= phase 3: normal = J^T g over this lane's levels, summed over the quarters ======================
        float nrm[3] = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < kE16LevelSteps; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float gv = gf[i >> 1][2 * (i & 1) + c];            // e16_grad_feature: tile i / 2, register 2 (i % 2) + c
#pragma unroll
                for (int d = 0; d < 3; ++d) nrm[d] += gv * jac_col[((i * 3 + d) * 2 + c) * 64];
            }
        const float geo3[3] = {o3[0][1], o3[0][2], o3[0][3]};
        float gsq = geo3[0] * geo3[0] + geo3[1] * geo3[1] + geo3[2] * geo3[2];
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
#pragma unroll
            for (int d = 0; d < 3; ++d) nrm[d] += __shfl_xor(nrm[d], off);
            gsq += __shfl_xor(gsq, off);
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) nrm[d] = nrm[d] / a.bound2;                         // d x01 / d xyz
        normalize_n<3>(nrm, 1e-10f);                                                    // renderer.py:192
        const float ginv = 1.0f / fmaxf(sqrtf(gsq), 1e-12f);                            // network.py:434-435 (F.normalize)
        const float head = o3[0][0];          // quarter 0: sdf, 1: roughness logit, 2: blend logit
        if (on) {
            if (a.geo) {
                const f32x3 gv = {geo3[0] * ginv, geo3[1] * ginv, geo3[2] * ginv};
                *reinterpret_cast<f32x3*>(a.geo + 12 * slot + 3 * quarter) = gv;
            }
            if (quarter == 0) {
                const float sgn = head > 0 ? 1.0f : (head < 0 ? -1.0f : 0.0f);
                const float sigma = a.inv_beta * (0.5f + 0.5f * sgn * expm1f(-fabsf(head) / a.beta)) * a.density_scale;
                if (a.alpha) a.alpha[slot] = 1.0f - expf(-sigma * a.dt[slot]);
                if (a.sigma) a.sigma[slot] = sigma;
            } else if (quarter == 1) {
                if (a.rough) a.rough[slot] = a.rough_act_scale * softplusf(head + a.rough_bias) * a.rough_scale;   // network.py:443-448
            } else if (quarter == 2) {
                if (a.blend) a.blend[slot] = head;
            } else if (a.normal) {
                const f32x3 nv = {nrm[0], nrm[1], nrm[2]};
                *reinterpret_cast<f32x3*>(a.normal + 3 * slot) = nv;
            }
        }
        wave_lds_sync();      // the parked Jacobian is rewritten by the next batch
    }
}

}  // namespace
