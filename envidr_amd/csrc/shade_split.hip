// Split-precision environment MLP (optional shading mode, never the default / headline path): the two evaluations of
// the environment network per sample -- IDE(rotated normal, kappa_diffuse) and IDE(reflected direction, roughness),
// network.py:524-541,586-600 -- on the fp16 matrix cores with every operand carried as a (hi, lo) fp16 pair
// (mlp_split.hip.h: up to 22-bit significands, fp32 accumulation).  The kernel writes the 2 x 12 normalised environment
// features of every sample; the heads then run in fp32 in k_shade_samples' PRE_ENV instantiation.
//
// One workgroup = 4 waves (one per SIMD) sharing the weight stream through LDS; one wave = 32 samples = 64 items:
// lanes 0-31 hold the normal-side item of sample n, lanes 32-63 the reflection-side item of the same sample, and the two
// groups of 32 items run through the network one after the other (the accumulators of one group fill the registers).
#include "fused_common.hip.h"
#include "mlp_split.hip.h"
#include "sh_core.hip.h"

using namespace envidr;

namespace {

constexpr uint32_t kSplitThreads = 256;

template <int IDE_DEG, int ENV_T>
struct EnvSplitLayout {
    static constexpr int TERMS = ide_terms(IDE_DEG), K1 = 2 * TERMS, S1 = (K1 + 15) / 16, SH = 2 * ENV_T;
    static constexpr int F1 = 0, F2 = F1 + split_layer_frags(S1, ENV_T), F3 = F2 + split_layer_frags(SH, ENV_T),
                         F4 = F3 + split_layer_frags(SH, ENV_T), Frags = F4 + split_layer_frags(SH, 1);
    // pass length in fragments: a multiple of the ring depth, and whole chunks past the resident part (mlp_split.hip.h)
    static_assert(kSplitAhead % kSplitChunkFrags == 0 && kSplitResident % kSplitChunkFrags == 0, "one padding serves both");
    static constexpr int Padded = (Frags + kSplitAhead - 1) / kSplitAhead * kSplitAhead;
    static_assert(Padded >= kSplitResident + kSplitSlots * kSplitChunkFrags, "a pass shorter than the resident part plus the ring");
    static constexpr int BiasTiles = 3 * ENV_T + 1;
};

template <int IDE_DEG, int ENV_T>
__global__ void __launch_bounds__(kSplitThreads, 1) k_env_split(const ShadeArgs a, const void* __restrict__ blob, const float* __restrict__ bias) {
    using L = EnvSplitLayout<IDE_DEG, ENV_T>;
    constexpr int TERMS = L::TERMS, S1 = L::S1, SH = L::SH;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __shared__ u32x4 s_w[kSplitLdsBytes / 16];
    __shared__ __attribute__((aligned(16))) float s_bias[L::BiasTiles * 32];
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (uint32_t i = threadIdx.x; i < (uint32_t)L::BiasTiles * 32; i += kSplitThreads) s_bias[i] = bias[i];
    __syncthreads();
    SplitFragRing<kSplitAhead> wp;
    wp.start(s_w, lane, wave, blob, L::Padded);
    const float* bias_lane = s_bias + (lane >> 5) * 16;
    auto bias_tile = [&](int tile) {
        f32x16 b;
        const float4* p = reinterpret_cast<const float4*>(bias_lane + tile * 32);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float4 v = p[q]; b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w; }
        return b;
    };
    uint32_t M = a.M;
    if (a.m_dev) { const uint32_t md = __builtin_amdgcn_readfirstlane(*a.m_dev); M = md > a.M ? 0u : md; }
    const uint32_t enc = lane >> 5;
    // every wave of a block runs the same number of rounds (the weight stream has block-wide barriers)
    for (uint32_t base = blockIdx.x * 128u; base < M; base += gridDim.x * 128u) {
        const uint32_t id = base + wave * 32u + (lane & 31u);
        const bool on = id < M;
        const size_t i = on ? id : 0;
        const size_t gi = a.slot ? (size_t)a.slot[i] : i;
        const size_t ray = a.ray_ids ? (size_t)a.ray_ids[i] : 0;
        const float* dir = a.ray_ids ? a.rays_d + 3 * ray : a.dirs + 3 * i;
        float nrm[3], vd[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { nrm[d] = on ? a.normals[3 * gi + d] : 0.0f; vd[d] = on ? dir[d] : 0.0f; }
        const float rough = a.roughness[(size_t)a.rough_stride * gi];
        // renderer.py:147-180 (the statements of k_shade_samples)
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
        // this lane's item: the normal-side (lower lanes) or the reflection-side (upper lanes) encoding of its sample
        float f[16 * S1];
#pragma unroll
        for (int k = 2 * TERMS; k < 16 * S1; ++k) f[k] = 0.0f;
        ide_eval<IDE_DEG, true>(enc ? wr[0] : nenv[0], enc ? wr[1] : nenv[1], enc ? wr[2] : nenv[2], enc ? rough : a.kappa_diffuse,
                          [&](int j, float re, float im) {
                              f[j] = re * a.light_scale;
                              f[TERMS + j] = im * a.light_scale;
                          });
        // B operands of the first layer (lane order: step j, half h, slot i = feature 16 j + 8 h + i) for the two groups
        half8 inh[2][S1], inl[2][S1];
#pragma unroll
        for (int j = 0; j < S1; ++j)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float u = f[16 * j + q], v = f[16 * j + 8 + q];
                swap_halves(u, v);            // u: group A (items of the lower lanes), v: group B (items of the upper lanes)
                _Float16 hi, lo;
                split_f16(u, hi, lo); inh[0][j][q] = hi; inl[0][j][q] = lo;
                split_f16(v, hi, lo); inh[1][j][q] = hi; inl[1][j][q] = lo;
            }
        f32x16 outA, outB;
#pragma unroll 1
        for (int grp = 0; grp < 2; ++grp) {
            half8 xh[S1], xl[S1];
#pragma unroll
            for (int j = 0; j < S1; ++j) { xh[j] = grp ? inh[1][j] : inh[0][j]; xl[j] = grp ? inl[1][j] : inl[0][j]; }
            half8 ph[SH], pl[SH], qh[SH], ql[SH];
            // every tile group's fp16 conversion runs in the MFMA gaps of the group after it (mlp_split.hip.h, "pending")
            f32x16 accA[kSplitGroup], accB[kSplitGroup], o;
            auto sink_p = [&](auto tc, auto jc, const f32x16& v) {
                constexpr int t = decltype(tc)::value, j = decltype(jc)::value;
                split_pair_to_step<j>(v, ph[2 * t + (j >= 4)], pl[2 * t + (j >= 4)]);
            };
            auto sink_q = [&](auto tc, auto jc, const f32x16& v) {
                constexpr int t = decltype(tc)::value, j = decltype(jc)::value;
                split_pair_to_step<j>(v, qh[2 * t + (j >= 4)], ql[2 * t + (j >= 4)]);
            };
            auto sink_none = [&](auto, auto, const f32x16&) {};
            split_layer_piped<S1, ENV_T, L::F1, L::Padded>(wp, xh, xl, [&](int t) { return bias_tile(t); }, sink_p, accA, accB, NoFiller{},
              [&](f32x16* c1, f32x16* o1, auto t0, auto gt) {
                const Drain<decltype(gt)::value, decltype(t0)::value, 2 * decltype(t0)::value, decltype(sink_p)> pend1{c1, &sink_p};
                split_layer_piped<SH, ENV_T, L::F2, L::Padded>(wp, ph, pl, [&](int t) { return bias_tile(ENV_T + t); }, sink_q, o1, c1, pend1,
                  [&](f32x16* c2, f32x16* o2, auto t0b, auto gtb) {
                    const Drain<decltype(gtb)::value, decltype(t0b)::value, 2 * decltype(t0b)::value, decltype(sink_q)> pend2{c2, &sink_q};
                    split_layer_piped<SH, ENV_T, L::F3, L::Padded>(wp, qh, ql, [&](int t) { return bias_tile(2 * ENV_T + t); }, sink_p, o2, c2, pend2,
                      [&](f32x16* c3, f32x16* o3, auto t0c, auto gtc) {
                        const Drain<decltype(gtc)::value, decltype(t0c)::value, 2 * decltype(t0c)::value, decltype(sink_p)> pend3{c3, &sink_p};
                        split_layer_piped<SH, 1, L::F4, L::Padded>(wp, ph, pl, [&](int) { return bias_tile(3 * ENV_T); }, sink_none, o3, c3, pend3,
                          [&](f32x16* c4, f32x16*, auto, auto) { o = c4[0]; });
                      });
                  });
              });
            wp.template end_pass<L::Frags, L::Padded>();
            if (grp == 0) outA = o; else outB = o;
        }
        float e[16];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float u = outA[r], v = outB[r];
            unpack_pair(u, v);
            e[tile_row(r, 0)] = u; e[tile_row(r, 1)] = v;     // rows 0-3, 8-11 and 4-7, 12-15 of this lane's own item
        }
        float e12[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) e12[k] = e[k];
        normalize_n<12>(e12, 1e-12f);                                               // network.py:541,600
        if (on) {
            float4* dst = reinterpret_cast<float4*>(a.env_pre + 24 * i + 12 * enc);
#pragma unroll
            for (int q = 0; q < 3; ++q) dst[q] = make_float4(e12[4 * q], e12[4 * q + 1], e12[4 * q + 2], e12[4 * q + 3]);
        }
    }
}

}  // namespace

namespace envidr {

int launch_env_split(const envidr_render_desc* d, const ShadeArgs& a, hipStream_t s, const char* who) {
    ENVIDR_REQUIRE(d->env_split_blob && d->env_split_bias && a.env_pre, "%s: split-precision mode without its weight blob / feature scratch", who);
    ENVIDR_REQUIRE(d->dir_sh_degree == 0, "%s: split precision belongs to the environment-MLP family", who);
    const uint32_t blocks = std::max(1u, std::min((uint32_t)device_cu_count(), ceil_div(a.M, 128u)));
    const dim3 grid(blocks), block(kSplitThreads);
    if (d->ide_degree == 5 && d->env_hidden == 256) hipLaunchKernelGGL((k_env_split<5, 8>), grid, block, 0, s, a, d->env_split_blob, d->env_split_bias);
    else if (d->ide_degree == 4 && d->env_hidden == 160) hipLaunchKernelGGL((k_env_split<4, 5>), grid, block, 0, s, a, d->env_split_blob, d->env_split_bias);
    else {
        set_error("%s: split precision is built for (ide_degree, env_hidden) = (5,256) and (4,160), not (%u,%u)", who, d->ide_degree, d->env_hidden);
        return ENVIDR_EINVAL;
    }
    return check_launch("k_env_split");
}

}  // namespace envidr

extern "C" {

uint32_t envidr_split_layer_halves(int k_order, uint32_t k_in, uint32_t m_out) {
    const SplitOrder o = k_order ? kSplitTileOrder : kSplitLaneOrder;
    return (uint32_t)split_layer_frags((int)split_steps_for(o, k_in), (int)(round_up(m_out, 32) / 32)) * kSplitFragHalves;
}
uint32_t envidr_split_group(void) { return (uint32_t)kSplitGroup; }
uint32_t envidr_split_chunk_bytes(void) { return kSplitBlobGrain; }

int envidr_pack_layer_split(const float* W_host, uint32_t m_out, uint32_t k_in, int k_order, uint16_t* dst_host) {
    ENVIDR_REQUIRE(W_host && dst_host && m_out && k_in, "pack_layer_split: null / empty argument");
    pack_split_weight(W_host, m_out, k_in, k_order ? kSplitTileOrder : kSplitLaneOrder, (uint32_t)kSplitGroup, dst_host);
    return ENVIDR_OK;
}

}  // extern "C"
