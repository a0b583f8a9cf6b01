// Split-precision environment MLP, fused-pair form (mlp_split2.hip.h): the two evaluations of the environment network per sample --
// IDE(rotated normal, kappa_diffuse) and IDE(reflected direction, roughness), network.py:524-541,586-600 -- on the fp16 matrix cores with
// every operand carried as a (hi, lo) fp16 pair.  Optional shading mode, never the default / headline path.  The kernel writes the 2 x 12
// normalised environment features of every sample; the heads then run in fp32 (k_shade_samples' PRE_ENV instantiation).
//
// One workgroup = 8 waves (two per SIMD, 256 registers each) sharing ONE weight stream through LDS; a round = 128 samples = 256 items, one
// pass over the weights: wave w takes the 32 samples of quarter w & 3 and the encoding w >> 2 (0: normal side, 1: reflection side).  The
// (hi, lo) layer-1 operands of a wave's 32 items wait in its own S1 x 2 KiB of LDS.
#include "fused_common.hip.h"
#include "mlp_split2.hip.h"
#include "sh_core.hip.h"

using namespace envidr;

namespace {

constexpr uint32_t kSplit2Threads = 64 * kS2Waves;
typedef uint16_t u16_alias __attribute__((may_alias));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_alias __attribute__((may_alias));

template <int IDE_DEG, int ENV_T>
__global__ void __launch_bounds__(kSplit2Threads, 1) k_env_split2(const ShadeArgs a, const void* __restrict__ blob, const float* __restrict__ bias) {
    constexpr int TERMS = ide_terms(IDE_DEG);
    using L = Split2Layout<TERMS, ENV_T>;
    constexpr int S1 = L::S1, SH = L::SH, FR = L::Padded;
    constexpr int NG = (ENV_T + 1) / 2;                       // layer-2 tiles are accumulated two at a time (the last group of an odd count: one)
    __shared__ u32x4 s_ring[kS2RingBytes / 16];
    __shared__ u32x4 s_in[kS2Waves * L::InFrags * 64];        // [wave][step][hi, lo][lane]
    __shared__ __attribute__((aligned(16))) float s_bias[L::BiasTiles * 32];
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (uint32_t i = threadIdx.x; i < (uint32_t)L::BiasTiles * 32; i += kSplit2Threads) s_bias[i] = bias[i];
    u32x4_alias* in_w = reinterpret_cast<u32x4_alias*>(s_in) + wave * (uint32_t)(L::InFrags * 64);       // this wave's operands
    // (the slots of features >= 2 TERMS stay zero for the whole kernel)
#pragma unroll
    for (int f = 0; f < L::InFrags; ++f) in_w[f * 64 + lane] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    Split2Ring<kS2Ahead> wp;
    wp.start(s_ring, lane, wave, blob, FR);
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
    const uint32_t* list = (a.list && (uint32_t)__builtin_amdgcn_readfirstlane(*a.m_all) != M) ? a.list : nullptr;
    const uint32_t enc = wave >> 2, quarter = wave & 3u;
    // Both lane halves evaluate the features of item n = lane & 31 (the matrix instruction wants them in lane n AND lane n + 32: slots
    // 8 h .. 8 h + 7 of a step live in half h); each writes the 16-bit slot of feature k at fragment (step k / 16, hi | lo), lane n + 32 h(k):
    // the two halves store the same bits to the same address.  Feature order = the first layer's column order, as in shade_split.hip.
    u16_alias* in_item = reinterpret_cast<u16_alias*>(in_w) + (lane & 31u) * 8u;
    // A round's record is fetched two rounds ahead in two steps, so that no round waits for memory: the indices (list -> slot, ray id) while
    // the round before last runs, the record itself (normal, view direction, roughness) through those indices while the last one runs.
    struct RecordIndex { uint32_t i, gi, ray; bool on; };
    struct RecordData { float nrm[3], vd[3], rough; };
    auto fetch_index = [&](uint32_t base) {
        const uint32_t id = base + quarter * 32u + (lane & 31u);
        RecordIndex r;
        r.on = id < M;                                          // (base itself may lie beyond M: a round that will not run)
        r.i = r.on ? (list ? list[id] : id) : 0u;
        r.gi = a.slot ? a.slot[r.i] : r.i;
        r.ray = a.ray_ids ? a.ray_ids[r.i] : 0u;
        return r;
    };
    auto fetch_data = [&](const RecordIndex& r) {
        RecordData d;
        const float* dir = a.ray_ids ? a.rays_d + 3 * (size_t)r.ray : a.dirs + 3 * (size_t)r.i;
#pragma unroll
        for (int k = 0; k < 3; ++k) { d.nrm[k] = r.on ? a.normals[3 * (size_t)r.gi + k] : 0.0f; d.vd[k] = r.on ? dir[k] : 0.0f; }
        d.rough = a.roughness[(size_t)a.rough_stride * r.gi];
        return d;
    };
    const uint32_t stride = gridDim.x * 128u, base0 = blockIdx.x * 128u;
    RecordIndex idx_now = fetch_index(base0), idx_next = fetch_index(base0 + stride);
    RecordData rec_next = fetch_data(idx_now);
    // every wave of a block runs the same number of rounds (the weight stream has block-wide barriers)
    for (uint32_t base = base0; base < M; base += stride) {
        const RecordData rec = rec_next;
        const bool on = idx_now.on;
        const size_t i = idx_now.i;
        idx_now = idx_next;
        rec_next = fetch_data(idx_now);                         // next round's record, through the indices fetched a round ago
        idx_next = fetch_index(base + 2u * stride);             // (a round that will not run: its ids lie beyond M, nothing is read)
        __builtin_amdgcn_sched_barrier(0);
        float nrm[3] = {rec.nrm[0], rec.nrm[1], rec.nrm[2]}, vd[3] = {rec.vd[0], rec.vd[1], rec.vd[2]};
        const float rough = rec.rough;
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
        ide_eval<IDE_DEG, true>(enc ? wr[0] : nenv[0], enc ? wr[1] : nenv[1], enc ? wr[2] : nenv[2], enc ? rough : a.kappa_diffuse,
                          [&](int j, float re, float im) {
                              _Float16 rh, rl, ih, il;
                              split_f16(re * a.light_scale, rh, rl);
                              split_f16(im * a.light_scale, ih, il);
                              auto at = [](int k) { return (uint32_t)((k / 16 * 2) * 512 + ((k >> 3) & 1) * 256 + (k & 7)); };
                              in_item[at(j)] = __builtin_bit_cast(uint16_t, rh);
                              in_item[at(j) + 512] = __builtin_bit_cast(uint16_t, rl);
                              in_item[at(TERMS + j)] = __builtin_bit_cast(uint16_t, ih);
                              in_item[at(TERMS + j) + 512] = __builtin_bit_cast(uint16_t, il);
                          });
        auto in_frag = [&](int s, int hl) { return __builtin_bit_cast(half8, (u32x4)in_w[(s * 2 + hl) * 64 + lane]); };

        // ---- phase A: layer 1 tile by tile, each tile consumed by all of layer 2 at once ------------------------------------------
        f32x16 acc2[ENV_T];
#pragma unroll
        for (int u = 0; u < ENV_T; ++u) acc2[u] = bias_tile(ENV_T + u);
        f32x16 acc1;
        half8 yh[2][2], yl[2][2];                             // [buffer][step of the tile]
        // (the accumulator already holds the tile's bias: loaded where the wait for it hides under MFMAs that do not need it)
        auto layer1_tile = [&](auto tc) {
            constexpr int t = decltype(tc)::value;
            half8 bh[2], bl[2];
            bh[0] = in_frag(0, 0); bl[0] = in_frag(0, 1);
            static_for<S1>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s + 1 < S1) { bh[(s + 1) & 1] = in_frag(s + 1, 0); bl[(s + 1) & 1] = in_frag(s + 1, 1); }
                split2_step<1, L::a1(t, s, 0), FR, 0, 1>(wp, bh[s & 1], bl[s & 1], &acc1, NoFill{});
            });
        };
        // accumulator pair j of the finished layer-1 tile -> (hi, lo) B operand of its two steps, into buffer BUF
        auto cvt1 = [&](auto bufc) {
            return [&](auto, auto jc) {
                constexpr int buf = decltype(bufc)::value, j = decltype(jc)::value;
                split_pair_to_step<j>(acc1, yh[buf][j >= 4], yl[buf][j >= 4]);
            };
        };
        acc1 = bias_tile(0);
        layer1_tile(std::integral_constant<int, 0>{});
        {
            const auto sink = cvt1(std::integral_constant<int, 0>{});
            CvtFill<1, decltype(sink)>{&sink}.template piece<0, 1>();
        }
        if constexpr (ENV_T > 1) { acc1 = bias_tile(1); layer1_tile(std::integral_constant<int, 1>{}); }
        static_for<ENV_T>([&](auto tc) {
            constexpr int t = decltype(tc)::value, buf = t & 1;
            // its two k-steps of every layer-2 tile; the next layer-1 tile is converted in the gaps of the first step, and the accumulator it
            // leaves takes the bias of the tile after next while the second step runs
            const auto sink = cvt1(std::integral_constant<int, (t + 1) & 1>{});
            const CvtFill<1, decltype(sink)> fill{&sink};
            constexpr int NQ = 3 * ENV_T;
            static_for<2>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s == 1 && t + 2 < ENV_T) acc1 = bias_tile(t + 2);
                static_for<NG>([&](auto pc) {
                    constexpr int p = decltype(pc)::value, GT = (ENV_T - 2 * p) < 2 ? (ENV_T - 2 * p) : 2, Q0 = 6 * p;
                    if constexpr (s == 0 && t + 1 < ENV_T) split2_step<GT, L::a2(t, s, 2 * p, 0), FR, Q0, NQ>(wp, yh[buf][s], yl[buf][s], &acc2[2 * p], fill);
                    else split2_step<GT, L::a2(t, s, 2 * p, 0), FR, 0, 1>(wp, yh[buf][s], yl[buf][s], &acc2[2 * p], NoFill{});
                });
            });
            if constexpr (t + 2 < ENV_T) layer1_tile(std::integral_constant<int, t + 2>{});
        });

        // ---- layer-2 output -> (hi, lo) in place; phase B: layer 3 tile by tile, each tile consumed by the last layer at once -----
        half8 xh[SH], xl[SH];
        auto cvt2_tile = [&](auto uc) {          // tile u -> steps 2 u, 2 u + 1
            return [&](auto, auto jc) {
                constexpr int u = decltype(uc)::value, j = decltype(jc)::value;
                split_pair_to_step<j>(acc2[u], xh[2 * u + (j >= 4)], xl[2 * u + (j >= 4)]);
            };
        };
        f32x16 acc3[2], acc4 = bias_tile(3 * ENV_T);
        acc3[0] = bias_tile(2 * ENV_T);
        if constexpr (ENV_T > 1) acc3[1] = bias_tile(2 * ENV_T + 1);
        {
            const auto sink = cvt2_tile(std::integral_constant<int, 0>{});
            CvtFill<1, decltype(sink)>{&sink}.template piece<0, 1>();
        }
        half8 zh[2], zl[2];                                   // the finished layer-3 tile's (hi, lo): [step]
        auto layer3_tile = [&](auto tc, const auto& fill_for_step) {
            constexpr int t = decltype(tc)::value;          // (acc3[t & 1] holds the tile's bias)
            static_for<SH>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                fill_for_step(sc, [&](const auto& fill, auto q0c, auto nqc) {
                    split2_step<1, L::b3(t, s, 0), FR, decltype(q0c)::value, decltype(nqc)::value>(wp, xh[s], xl[s], &acc3[t & 1], fill);
                });
            });
        };
        // tile 0 of layer 3: the remaining layer-2 tiles are converted two steps ahead of the step that reads them
        layer3_tile(std::integral_constant<int, 0>{}, [&](auto sc, auto&& run) {
            constexpr int s = decltype(sc)::value, u = s / 2 + 1;
            if constexpr (u < ENV_T) {
                const auto sink = cvt2_tile(std::integral_constant<int, u>{});
                const CvtFill<1, decltype(sink)> fill{&sink};
                run(fill, std::integral_constant<int, 3 * (s & 1)>{}, std::integral_constant<int, 6>{});
            } else run(NoFill{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        });
        static_for<ENV_T>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const auto sink = [&](auto, auto jc) {
                constexpr int j = decltype(jc)::value;
                split_pair_to_step<j>(acc3[t & 1], zh[j >= 4], zl[j >= 4]);
            };
            const CvtFill<1, decltype(sink)> fill{&sink};
            if constexpr (t + 1 < ENV_T) {
                layer3_tile(std::integral_constant<int, t + 1>{}, [&](auto sc, auto&& run) {
                    run(fill, std::integral_constant<int, 3 * decltype(sc)::value>{}, std::integral_constant<int, 3 * SH>{});
                });
            } else fill.template piece<0, 1>();
            if constexpr (t + 2 < ENV_T) acc3[t & 1] = bias_tile(2 * ENV_T + t + 2);          // converted; its wait hides under the last layer's MFMAs
            static_for<2>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                split2_step<1, L::b4(t, s, 0), FR, 0, 1>(wp, zh[s], zl[s], &acc4, NoFill{});
            });
        });
        wp.template end_pass<L::Frags, FR>();

        // rows tile_row(r, h) of item n are in lane n + 32 h: rows 0-3, 8-11 in the lower half (registers 0-7), rows 4-7 in registers 0-3 of the
        // upper.  Both halves collect the twelve features (so that the norm is summed in feature order, as everywhere else) and store their own.
        float e12[12];
        const bool lower = lane < 32u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = acc4[r], y = acc4[4 + r];
            const float ox = __shfl_xor(x, 32), oy = __shfl_xor(y, 32);
            e12[r] = lower ? x : ox;
            e12[4 + r] = lower ? ox : x;
            e12[8 + r] = lower ? y : oy;
        }
        normalize_n<12>(e12, 1e-12f);                                               // network.py:541,600
        if (on) {
            float* dst = a.env_pre + 24 * i + 12 * enc;
            if (lower) {
                *reinterpret_cast<float4*>(dst) = make_float4(e12[0], e12[1], e12[2], e12[3]);
                *reinterpret_cast<float4*>(dst + 8) = make_float4(e12[8], e12[9], e12[10], e12[11]);
            } else *reinterpret_cast<float4*>(dst + 4) = make_float4(e12[4], e12[5], e12[6], e12[7]);
        }
    }
}

}  // namespace

namespace envidr {

int launch_env_split2(const envidr_render_desc* d, const ShadeArgs& a, hipStream_t s, const char* who) {
    ENVIDR_REQUIRE(d->env_split_blob && d->env_split_bias && a.env_pre, "%s: split-precision mode without its weight blob / feature scratch", who);
    ENVIDR_REQUIRE(d->dir_sh_degree == 0, "%s: split precision belongs to the environment-MLP family", who);
    const uint32_t blocks = std::max(1u, std::min((uint32_t)device_cu_count(), ceil_div(a.M, 128u)));
    const dim3 grid(blocks), block(kSplit2Threads);
    if (d->ide_degree == 5 && d->env_hidden == 256) hipLaunchKernelGGL((k_env_split2<5, 8>), grid, block, 0, s, a, d->env_split_blob, d->env_split_bias);
    else if (d->ide_degree == 4 && d->env_hidden == 160) hipLaunchKernelGGL((k_env_split2<4, 5>), grid, block, 0, s, a, d->env_split_blob, d->env_split_bias);
    else {
        set_error("%s: split precision is built for (ide_degree, env_hidden) = (5,256) and (4,160), not (%u,%u)", who, d->ide_degree, d->env_hidden);
        return ENVIDR_EINVAL;
    }
    return check_launch("k_env_split2");
}

}  // namespace envidr

extern "C" {

// halves of the weight blob of the fused-pair kernel for (ide_degree, env_hidden) = (5, 256) or (4, 160); 0 for any other shape
uint32_t envidr_env_split2_halves(uint32_t ide_degree, uint32_t env_hidden) {
    if (ide_degree == 5 && env_hidden == 256) return (uint32_t)Split2Layout<ide_terms(5), 8>::Padded * kSplitFragHalves;
    if (ide_degree == 4 && env_hidden == 160) return (uint32_t)Split2Layout<ide_terms(4), 5>::Padded * kSplitFragHalves;
    return 0;
}

// W1 [H, 2 TERMS], W2, W3 [H, H], W4 [12, H] (row-major fp32, host) -> the (hi, lo) fp16 fragments in the kernel's consumption order
int envidr_pack_env_split2(const float* W1, const float* W2, const float* W3, const float* W4, uint32_t ide_degree, uint32_t env_hidden, uint16_t* dst_host) {
    ENVIDR_REQUIRE(W1 && W2 && W3 && W4 && dst_host, "pack_env_split2: null argument");
    if (ide_degree == 5 && env_hidden == 256) pack_env_split2<ide_terms(5), 8>(W1, W2, W3, W4, dst_host);
    else if (ide_degree == 4 && env_hidden == 160) pack_env_split2<ide_terms(4), 5>(W1, W2, W3, W4, dst_host);
    else {
        set_error("pack_env_split2: built for (ide_degree, env_hidden) = (5,256) and (4,160), not (%u,%u)", ide_degree, env_hidden);
        return ENVIDR_EINVAL;
    }
    return ENVIDR_OK;
}

}  // extern "C"
