// One pass of the environment MLP  2 TERMS -> 32 ENV_T -> 32 ENV_T -> 32 ENV_T -> 12  (network.py:527-546, 588-607) for a
// group of 32 samples, as the shading kernels run it -- 94 % of a sample's FLOPs, so every cycle of it that is not an MFMA
// issue slot is accounted for here:
//
//   * biases   : the blob carries each layer's bias as leading fragments (envidr_pack_layer).  For the three wide layers the
//                kernel copies them into LDS once (env_lds_init) and the accumulators START at the bias: 4 ds_read_b128 per
//                tile, issued between the MFMAs of the layer before, instead of one MFMA per output tile and layer spent on
//                1 * bias.  The bias fragments are skipped in the weight stream (a load without an MFMA).
//   * ReLU     : a wave's own vector-ALU instructions stall its matrix pipe (tools/probe/valu_overlap_probe.hip), LDS
//                instructions do not.  The 16 B operands of input tile K+1 go  accumulator --ds_max_f32 on a zero slot--> LDS
//                --ds_read--> VGPR  between tile K's MFMAs (mlp_mfma.hip.h "ReLU in the LDS atomic unit"); only a layer's
//                first tile is staged with v_accvgpr_read + v_max ahead of its MFMAs.
//   * 12 outputs on 16-row MFMA blocks (pipe_layer16_from_tiles).
// tools/probe/env_pass_probe.hip times exactly this function.
#pragma once
#include "mlp_mfma.hip.h"

namespace envidr {

template <int TERMS, int ENV_T>
struct EnvLayout {
    static constexpr int E1 = 0, E2 = E1 + lane_layer_frags(TERMS, ENV_T, true), E3 = E2 + tile_layer_frags(ENV_T, ENV_T, true),
                         E4 = E3 + tile_layer_frags(ENV_T, ENV_T, true), Frags = E4 + tile_layer_frags(ENV_T, 1, true);
    // per-wave LDS: the ReLU staging slot, then the bias tiles of E1, E2, E3 as packed row vectors ([tile][half][16])
    static constexpr int kBiasFloats = 3 * ENV_T * 32;
    static constexpr int kLdsFloats = kLdsStageFloats + kBiasFloats;
};

struct EnvAux {
    float* slot;          // LDS: this lane's column of the wave's staging slot
    const float* bias;    // LDS: bias tiles + this lane half's 16 floats
};

// once per wave: bias fragments of the blob (lane l < 32 of fragment t holds bias[32 t + l]) -> row-vector tiles in LDS
// (register r of lane half h holds feature tile_row(r, h)); staging slot zeroed.  `lds`: the wave's kLdsFloats floats.
template <int TERMS, int ENV_T>
__device__ __forceinline__ EnvAux env_lds_init(const float* __restrict__ env_blob, float* lds, uint32_t lane) {
    using L = EnvLayout<TERMS, ENV_T>;
    constexpr int first[3] = {L::E1, L::E2, L::E3};
    const uint32_t h = (lane >> 2) & 1u, r = (lane & 3u) + 4u * (lane >> 3);       // inverse of tile_row for lane < 32
#pragma unroll
    for (int layer = 0; layer < 3; ++layer)
#pragma unroll
        for (int t = 0; t < ENV_T; ++t) {
            const float v = env_blob[(size_t)(first[layer] + t) * 64 + lane];
            if (lane < 32) lds[kLdsStageFloats + (layer * ENV_T + t) * 32 + h * 16 + r] = v;
        }
#pragma unroll
    for (int i = 0; i < 16; ++i) lds[i * 64 + lane] = 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return EnvAux{lds + lane, lds + kLdsStageFloats + (lane >> 5) * 16};
}

// accumulator tile <- bias tile `tile` (of all three layers' tiles, counted from E1's first)
__device__ __forceinline__ f32x16 lds_bias_tile(const float* bias_half, int tile) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4* p = reinterpret_cast<const f32x4*>(bias_half + tile * 32);
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 w = p[q];
        v[4 * q] = w[0]; v[4 * q + 1] = w[1]; v[4 * q + 2] = w[2]; v[4 * q + 3] = w[3];
    }
    return v;
}

template <int I, int N, int FRAGS, typename Src>
__device__ __forceinline__ void skip_frags(Src& wp) {
    if constexpr (N > 0) {
        (void)wp.template take<I, FRAGS>();
        skip_frags<I + 1, N - 1, FRAGS>(wp);
    }
}

// E1: input = per-lane packed IDE terms; acc holds the bias; `other` (the next layer's accumulators) is loaded meanwhile
template <int S, int STEPS, int MT, int W0, int FRAGS, typename Src>
__device__ __forceinline__ void env_step_lanes(Src& wp, const float (&in)[STEPS], f32x16 (&acc)[MT], f32x16 (&other)[MT], const float* bias, int other_tile0) {
    pipe_one_step<MT, W0 + S * MT, FRAGS>(wp, acc, in[S]);
    if constexpr (S >= 2 && S - 2 < MT) other[S - 2 < MT ? S - 2 : 0] = lds_bias_tile(bias, other_tile0 + S - 2);
    __builtin_amdgcn_sched_barrier(0);
}
// (ds_max_f32 follows IEEE maxNum: max(NaN, 0) = 0.  A NaN pre-activation therefore becomes 0 here where torch.relu would propagate it;
//  finite inputs -- everything the parity tests and the reference's own frames contain -- are unaffected.)
// Piece J (0 .. 31) of the LDS round trip that stages tile KN's 16 operands: 16 ds_max_f32 (accumulator -> zero slot),
// then 8 x (read two operands), then 8 x (zero two words).  LDS executes a wave's instructions in order, so the pieces only
// have to be ISSUED in this order; one piece rides behind each MFMA (clumps of 16 overran the 64-cycle shadow of one MFMA).
template <int J, int KN, int KT>
__device__ __forceinline__ void lds_stage_piece(const f32x16 (&in)[KT], float* slot, float (&bq)[16]) {
    if constexpr (J < 16) {
        __hip_atomic_fetch_max(slot + J * 64, in[KN][J], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if constexpr (J < 24) {
        constexpr int r = 2 * (J - 16);
        bq[r] = slot[r * 64];
        bq[r + 1] = slot[(r + 1) * 64];
    } else if constexpr (J < 32) {
        constexpr int r = 2 * (J - 24);
        slot[r * 64] = 0.0f;
        slot[(r + 1) * 64] = 0.0f;
    }
}

// E2 / E3: input tiles `in` (dead one by one as their operands are staged), acc holds the bias.  NEXT_BIAS: in[K] is
// reloaded with bias tile next_tile0 + K -- it is the accumulator of the layer after this one.
template <int K, int S, int KT, int MT, int W0, int FRAGS, bool NEXT_BIAS, typename Src>
__device__ __forceinline__ void env_step_tiles(Src& wp, f32x16 (&in)[KT], f32x16 (&acc)[MT], float (&bq)[2][16], const EnvAux& aux, int next_tile0) {
    constexpr int FI = W0 + (K * 16 + S) * MT;
    [&]<int... T>(std::integer_sequence<int, T...>) {
        ((acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(wp.template take<FI + T, FRAGS>(), bq[K & 1][S], acc[T], 0, 0, 0),
          (K + 1 < KT && S * MT + T < 32 ? lds_stage_piece<(S * MT + T < 32 ? S * MT + T : 32), (K + 1 < KT ? K + 1 : K)>(in, aux.slot, bq[(K + 1) & 1]) : void()),
          __builtin_amdgcn_sched_barrier(0)), ...);
    }(std::make_integer_sequence<int, MT>{});
    if constexpr (NEXT_BIAS && S == 8) {
        in[K] = lds_bias_tile(aux.bias, next_tile0 + K);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// E4: 16-row blocks, two alternating accumulators; two staging pieces per (32-cycle) MFMA
template <int K, int S, int KT, int W0, int FRAGS, typename Src>
__device__ __forceinline__ void env_step_out16(Src& wp, const f32x16 (&in)[KT], f32x16& a0, f32x16& a1, float (&bq)[2][16], const EnvAux& aux) {
    if constexpr (S % 2 == 0) a0 = __builtin_amdgcn_mfma_f32_16x16x1f32(wp.template take<W0 + K * 16 + S, FRAGS>(), bq[K & 1][S], a0, 0, 0, 0);
    else a1 = __builtin_amdgcn_mfma_f32_16x16x1f32(wp.template take<W0 + K * 16 + S, FRAGS>(), bq[K & 1][S], a1, 0, 0, 0);
    if constexpr (K + 1 < KT) {
        lds_stage_piece<2 * S, (K + 1 < KT ? K + 1 : K)>(in, aux.slot, bq[(K + 1) & 1]);
        lds_stage_piece<2 * S + 1, (K + 1 < KT ? K + 1 : K)>(in, aux.slot, bq[(K + 1) & 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// which widths run the hand-over form (an even number of 32-feature tiles); the host packs E1 with k_order 4 and E2, E3 with 3 for
// exactly these (envidr_amd/fused.py: env_orders)
// (E1 loads the next layer's bias tiles behind its step-major steps 2 .. ENV_T + 1: with the hand-over its first TERMS - 16 steps must cover them)
constexpr bool env_handoff(int terms, int env_t) { return env_t >= 4 && env_t % 2 == 0 && terms - 16 >= env_t + 2; }

// ---- hand-over of a layer's first input tile ---------------------------------------------------------------------------
// The operands of a layer's FIRST input tile have nothing of that layer to hide under; staged with the vector ALU they sit in
// front of its MFMAs (174 cycles, three times per pass).  The layer BEFORE can stage them if its output tile 0 is finished early:
// its last input tile (E1: its last 16 steps) runs tile-major -- all 16 steps of output tile 0, then of tile 1 ... (the packed
// fragments of that stretch are laid out in this order, envidr_pack_layer k_order 3 / 4) -- and while tiles 1 .. MT-1 are being
// finished, tile 0 goes through the LDS round trip into bq[0], which the last input tile (odd index) does not use.
template <int T, int S, int MT, int W0, int FRAGS, bool NEXT_BIAS, int KT, typename Src>
__device__ __forceinline__ void env_step_tail(Src& wp, f32x16 (&in)[KT], f32x16 (&acc)[MT], const float bv, float (&bq0)[16], const EnvAux& aux, int next_tile0) {
    acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(wp.template take<W0 + T * 16 + S, FRAGS>(), bv, acc[T], 0, 0, 0);
    constexpr int J = (T - 1) * 16 + S;            // pieces ride behind the MFMAs of tiles 1 and 2
    if constexpr (T >= 1 && J < 32) lds_stage_piece<(J >= 0 && J < 32 ? J : 32), 0>(acc, aux.slot, bq0);
    if constexpr (NEXT_BIAS && T == MT / 2 && S == 0) in[KT - 1] = lds_bias_tile(aux.bias, next_tile0 + KT - 1);
    __builtin_amdgcn_sched_barrier(0);
}

// FRAGS: the pass's fragment count padded to the weight ring's depth.  out16: see pipe_layer16_from_tiles / fold16.
template <int TERMS, int ENV_T, int FRAGS, bool HANDOFF = false, typename Src>
__device__ __forceinline__ void env_pass(Src& wp, const uint32_t lane, const EnvAux& aux, const float (&in)[TERMS], f32x16& out16) {
    using L = EnvLayout<TERMS, ENV_T>;
    f32x16 ha[ENV_T], hb[ENV_T];
    float bq[2][16];
    static_assert(!HANDOFF || env_handoff(TERMS, ENV_T), "hand-over needs an even tile count (the last tile uses bq[1]) and TERMS - 16 >= ENV_T + 2 step-major steps in E1");
    // ---- E1: 2 TERMS -> 32 ENV_T
#pragma unroll
    for (int t = 0; t < ENV_T; ++t) ha[t] = lds_bias_tile(aux.bias, t);
    skip_frags<L::E1, ENV_T, FRAGS>(wp);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int kHead = HANDOFF ? TERMS - 16 : TERMS;        // E1's step-major steps
    [&]<int... S>(std::integer_sequence<int, S...>) {
        (env_step_lanes<S, TERMS, ENV_T, L::E1 + ENV_T, FRAGS>(wp, in, ha, hb, aux.bias, ENV_T), ...);
    }(std::make_integer_sequence<int, kHead>{});
    if constexpr (HANDOFF) {
        [&]<int... TS>(std::integer_sequence<int, TS...>) {
            (env_step_tail<TS / 16, TS % 16, ENV_T, L::E1 + ENV_T + kHead * ENV_T, FRAGS, false>(wp, hb, ha, in[kHead + TS % 16], bq[0], aux, 0), ...);
        }(std::make_integer_sequence<int, 16 * ENV_T>{});
    }
    // ---- E2: ha -> hb, ha reloaded with E3's bias
    skip_frags<L::E2, ENV_T, FRAGS>(wp);
    if constexpr (!HANDOFF) stage_operands<0, 0, 16, true>(ha, bq[0]);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int kTiles = HANDOFF ? ENV_T - 1 : ENV_T;        // step-major input tiles of E2 / E3
    [&]<int... KS>(std::integer_sequence<int, KS...>) {
        (env_step_tiles<KS / 16, KS % 16, ENV_T, ENV_T, L::E2 + ENV_T, FRAGS, true>(wp, ha, hb, bq, aux, 2 * ENV_T), ...);
    }(std::make_integer_sequence<int, 16 * kTiles>{});
    if constexpr (HANDOFF) {
        [&]<int... TS>(std::integer_sequence<int, TS...>) {
            (env_step_tail<TS / 16, TS % 16, ENV_T, L::E2 + ENV_T + kTiles * 16 * ENV_T, FRAGS, true>(wp, ha, hb, bq[1][TS % 16], bq[0], aux, 2 * ENV_T), ...);
        }(std::make_integer_sequence<int, 16 * ENV_T>{});
    }
    // ---- E3: hb -> ha
    skip_frags<L::E3, ENV_T, FRAGS>(wp);
    if constexpr (!HANDOFF) stage_operands<0, 0, 16, true>(hb, bq[0]);
    __builtin_amdgcn_sched_barrier(0);
    [&]<int... KS>(std::integer_sequence<int, KS...>) {
        (env_step_tiles<KS / 16, KS % 16, ENV_T, ENV_T, L::E3 + ENV_T, FRAGS, false>(wp, hb, ha, bq, aux, 0), ...);
    }(std::make_integer_sequence<int, 16 * kTiles>{});
    if constexpr (HANDOFF) {
        [&]<int... TS>(std::integer_sequence<int, TS...>) {
            (env_step_tail<TS / 16, TS % 16, ENV_T, L::E3 + ENV_T + kTiles * 16 * ENV_T, FRAGS, false>(wp, hb, ha, bq[1][TS % 16], bq[0], aux, 0), ...);
        }(std::make_integer_sequence<int, 16 * ENV_T>{});
    }
    // ---- E4: ha -> 12 outputs (bias through the stream: one 32-cycle MFMA)
    f32x16 a0, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.0f; a1[r] = 0.0f; }
    a1 = __builtin_amdgcn_mfma_f32_16x16x1f32(wp.template take<L::E4, FRAGS>(), lane < 32 ? 1.0f : 0.0f, a1, 0, 0, 0);
    if constexpr (!HANDOFF) stage_operands<0, 0, 16, true>(ha, bq[0]);
    __builtin_amdgcn_sched_barrier(0);
    [&]<int... KS>(std::integer_sequence<int, KS...>) {
        (env_step_out16<KS / 16, KS % 16, ENV_T, L::E4 + 1, FRAGS>(wp, ha, a0, a1, bq, aux), ...);
    }(std::make_integer_sequence<int, 16 * ENV_T>{});
#pragma unroll
    for (int r = 0; r < 16; ++r) out16[r] = a0[r] + a1[r];
    wp.template end_pass<L::Frags>();
}

}  // namespace envidr
