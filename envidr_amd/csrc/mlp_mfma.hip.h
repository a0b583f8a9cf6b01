// Register-resident tiny-MLP evaluation on the gfx950 matrix cores (exact-fp32 MFMA).
//
// One wave64 pushes 32 samples ("a group") at a time through a chain of small dense layers.
// The contraction is v_mfma_f32_32x32x2_f32:  D[32 x 32] += A[32 x 2] * B[2 x 32]  with
//   A = a 32-output-feature tile of the layer's weights  (lane l holds W[m0 + (l & 31)][k(l >> 5)])
//   B = activations, feature-major                        (lane l holds X[k(l >> 5)][sample l & 31])
//   D = lane l, register r: feature m0 + (r & 3) + 8 (r >> 2) + 4 (l >> 5) of sample (l & 31).
//
// The trick that keeps a whole MLP in registers: a D tile has exactly the shape a B operand needs
// (lanes <-> samples, the two lane halves hold two different features).  The reduction index k of the
// next layer may be visited in ANY order as long as A and B agree, so step s = 16 T + r of the next
// layer simply takes accumulator register r of tile T as its B operand -- no LDS, no shuffles, no
// transposes between layers.  The weights are pre-permuted on the host (pack_linear below) so that the
// A fragment of every (step, output tile) is one contiguous 256-byte row: one coalesced
// global_load_dword per MFMA, streamed from L2 (all layers of the ENVIDR shading networks together
// are < 1 MB, L2-resident on every XCD).
//
// Numerics: each output is an fp32 FMA chain over k in the packed order (MFMA f32 is bitwise an
// fmaf chain, MI355X_MICROARCH.md), starting from the bias.  Parity against torch's fp32 GEMMs is
// fp32-rounding-level (tests/test_fused_gpu.py).
#pragma once
#include "common.hip.h"
#include <utility>

namespace envidr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// feature index (inside a 32-feature tile) held by accumulator register r in lane half h
__host__ __device__ constexpr int tile_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- input orderings -----------------------------------------------------------------------
// kLaneOrder : the layer input comes from per-lane feature registers packed with pack_pair();
//              step s uses features (2s, 2s+1) in lane halves (0, 1).
// kTileOrder : the layer input is the previous layer's accumulator tiles;
//              step s = 16 T + r uses feature 32 T + tile_row(r, h).
enum KOrder : int { kLaneOrder = 0, kTileOrder = 1 };

__host__ __device__ constexpr int k_of_step(KOrder order, int s, int h) {
    return order == kLaneOrder ? 2 * s + h : 32 * (s >> 4) + tile_row(s & 15, h);
}
constexpr uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }
constexpr uint32_t steps_for(KOrder order, uint32_t k_in) { return order == kLaneOrder ? (k_in + 1) / 2 : round_up(k_in, 32) / 2; }
// with_bias: the layer's bias travels in the weight stream as one extra reduction step placed FIRST
// (A = bias in lane half 0, zero in half 1; B = 1 in half 0): acc = 0 + 1 * bias, exactly the value a
// separate bias load would initialise the accumulator with, but prefetched like any other fragment
// instead of being a synchronous load at the head of every layer.
constexpr uint32_t packed_weight_floats(KOrder order, uint32_t k_in, uint32_t m_out, bool with_bias = false) {
    return (steps_for(order, k_in) + (with_bias ? 1u : 0u)) * (round_up(m_out, 32) / 32) * 64;
}
constexpr uint32_t packed_bias_floats(uint32_t m_out) { return round_up(m_out, 32); }

// Host-side packing.  W is row-major [m_out][k_in] (torch nn.Linear.weight); `transpose` packs W^T
// instead (used for the input-gradient layers).  dst: [step][m_tile][64].
inline void pack_linear(const float* W, uint32_t m_out, uint32_t k_in, bool transpose, KOrder order, float* dst,
                        const float* bias = nullptr) {
    const uint32_t M = transpose ? k_in : m_out, K = transpose ? m_out : k_in;   // logical layer: K -> M
    const uint32_t steps = steps_for(order, K), mt = round_up(M, 32) / 32;
    if (bias) {
        for (uint32_t t = 0; t < mt; ++t)
            for (uint32_t lane = 0; lane < 64; ++lane) {
                const uint32_t m = 32 * t + (lane & 31);
                dst[(size_t)t * 64 + lane] = (lane < 32 && m < M) ? bias[m] : 0.0f;
            }
        dst += (size_t)mt * 64;
    }
    for (uint32_t s = 0; s < steps; ++s)
        for (uint32_t t = 0; t < mt; ++t)
            for (uint32_t lane = 0; lane < 64; ++lane) {
                const uint32_t h = lane >> 5, i = lane & 31;
                const uint32_t m = 32 * t + i, k = (uint32_t)k_of_step(order, (int)s, (int)h);
                float v = 0.0f;
                if (m < M && k < K) v = transpose ? W[(size_t)k * k_in + m] : W[(size_t)m * k_in + k];
                dst[((size_t)s * mt + t) * 64 + lane] = v;
            }
}
// per-feature vector (bias, or a weight row) in accumulator layout: dst[tile][half][reg]
inline void pack_rowvec(const float* v, uint32_t m_out, float* dst) {
    const uint32_t mt = round_up(m_out, 32) / 32;
    for (uint32_t t = 0; t < mt; ++t)
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) {
                const uint32_t m = 32 * t + (uint32_t)tile_row(r, h);
                dst[t * 32 + h * 16 + r] = m < m_out ? v[m] : 0.0f;
            }
}

// The LAST 16 reduction steps of a packed layer in tile-major order ([tile][step] instead of [step][tile]): the environment pass
// finishes output tile 0 first there and stages it for the next layer while the other tiles are completed (env_pass.hip.h
// "hand-over"; k_order 3 = tile order + this, 4 = lane order + this).  `w`: the layer's weight fragments (behind the bias step).
inline void retile_tail(float* w, uint32_t steps, uint32_t mt) {
    if (steps < 16) return;
    const uint32_t s0 = steps - 16;
    float* tail = w + (size_t)s0 * mt * 64;
    float* tmp = new float[(size_t)16 * mt * 64];
    memcpy(tmp, tail, sizeof(float) * 16 * mt * 64);
    for (uint32_t t = 0; t < mt; ++t)
        for (uint32_t q = 0; q < 16; ++q) memcpy(tail + ((size_t)t * 16 + q) * 64, tmp + ((size_t)q * mt + t) * 64, sizeof(float) * 64);
    delete[] tmp;
}

// ---- layers with at most 16 outputs ------------------------------------------------------------
// A layer like env 256 -> 12 or a head's 64 -> 3 fills 12 (3) of a 32-feature tile's rows.  It runs on
// v_mfma_f32_16x16x1_4B_f32 instead (four independent 16x16x1 blocks in one 32-cycle instruction; lane l belongs to
// block l >> 4 and holds one A entry (row l & 15) and one B entry (column l & 15) of it).  Accumulator register r of a
// 32x32 input tile, read as that instruction's B operand, IS four such blocks:
//     block 0: feature tile_row(r, 0) of samples  0-15      block 1: the same feature of samples 16-31
//     block 2: feature tile_row(r, 1) of samples  0-15      block 3: the same feature of samples 16-31
// so with A = W[l & 15][the feature the lane's block sees] the 16 outputs of 32 samples cost one 32-cycle MFMA per input
// register -- half the time of a 32x32x2 step.  D register 4 b + q of lane l is block b's row 4 (l >> 4) + q, column l & 15;
// a sample's output is the sum of its two blocks (fold16).  Fragment = 64 floats per reduction step as for a one-tile layer,
// so the pass blobs keep their sizes; k_order 2 in envidr_pack_layer.
inline void pack_linear16(const float* W, uint32_t m_out, uint32_t k_in, float* dst, const float* bias = nullptr) {
    const uint32_t steps = steps_for(kTileOrder, k_in);
    if (bias) {
        for (uint32_t lane = 0; lane < 64; ++lane) dst[lane] = (lane < 32 && (lane & 15) < m_out) ? bias[lane & 15] : 0.0f;
        dst += 64;
    }
    for (uint32_t s = 0; s < steps; ++s)
        for (uint32_t lane = 0; lane < 64; ++lane) {
            const uint32_t i = lane & 15, k = (uint32_t)k_of_step(kTileOrder, (int)s, (int)(lane >> 5));
            dst[(size_t)s * 64 + lane] = (i < m_out && k < k_in) ? W[(size_t)i * k_in + k] : 0.0f;
        }
}

// ---- device side ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Half exchange across the wave (v_permlane32_swap): afterwards
//   a = [a.lower | b.lower],  b = [a.upper | b.upper]      (".lower" = lanes 0-31)
__device__ __forceinline__ void swap_halves(float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
// per-lane features (even, odd) of the lane's own sample  ->  B operands of step s for group A
// (samples of lanes 0-31, returned in `even`) and group B (samples of lanes 32-63, in `odd`).
__device__ __forceinline__ void pack_pair(float& even, float& odd) { swap_halves(even, odd); }
// accumulator register r of group A's tile and of group B's tile  ->  for every lane, its own
// sample's features tile_row(r, 0) (returned in `a`) and tile_row(r, 1) (returned in `b`).
__device__ __forceinline__ void unpack_pair(float& a, float& b) { swap_halves(a, b); }

// ---- weight / bias fetch --------------------------------------------------------------------
// All parameter reads go through a buffer resource (SGPR descriptor) with
//   voffset = lane * 4 (one VGPR for the whole kernel), soffset = wave-uniform byte offset, imm offset.
// Flat/global addressing would need a 64-bit per-lane VGPR address for every 4 KiB window of
// weights; the compiler hoists those out of the persistent loop by the thousand and spills them.
struct ParamBuf {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t lane_off;    // lane * 4
    uint32_t half16;      // (lane >> 5) * 64: byte offset of this lane half's 16 floats inside a packed row-vector tile
};
// `bytes` is the exact size of the parameter array: buffer loads beyond it return 0 instead of
// faulting, which lets the weight stream prefetch past the end of a layer.
__device__ __forceinline__ ParamBuf make_param_buf(const float* base, uint32_t bytes, uint32_t lane) {
    ParamBuf p;
    // wave-uniform pointer (kernel argument): descriptor lives in SGPRs; 0x00020000 = raw dword buffer
    p.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
    p.lane_off = lane * 4u;
    p.half16 = (lane >> 5) * 64u;
    return p;
}
__device__ __forceinline__ float param_load(const ParamBuf& p, uint32_t byte_off_uniform) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(p.rsrc, p.lane_off, byte_off_uniform, 0));
}

// acc tile <- packed row vector (bias or a weight row), tile `tile`
__device__ __forceinline__ f32x16 load_rowvec(const ParamBuf& p, int tile) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(p.rsrc, p.half16, (uint32_t)(tile * 128 + q * 16), 0);
        v[4 * q] = __uint_as_float(w[0]); v[4 * q + 1] = __uint_as_float(w[1]);
        v[4 * q + 2] = __uint_as_float(w[2]); v[4 * q + 3] = __uint_as_float(w[3]);
    }
    return v;
}

// =============================================================================================
// Weight stream
// =============================================================================================
// Weights are laid out in memory in consumption order ("pass blobs": the layers of a pass concatenated, zero-padded to
// whole 16 KiB chunks of 64 fragments -- fused.py `blob`), so a pass is one linear stream of 256-byte fragments.
constexpr int kChunkFrags = 64;
constexpr int kChunkFloats = kChunkFrags * 64;      // 4096 floats = 16 KiB
constexpr uint32_t kChunkBytes = kChunkFloats * 4;

// Every wave streams the pass blobs itself from L2 through a ring of PF registers
// (fragment I lives in ring[I % PF]; taking it re-issues the load of fragment I + PF).  The ring runs
// CONTINUOUSLY across layers, passes and rounds -- at the end of a pass it rolls over into the next
// pass's blob -- so the L2 latency is exposed once per kernel, not once per layer.  Requires every
// pass to consume a multiple of PF fragments (end_pass pads with dummy takes).
template <int PF>
struct WeightRing {
    __amdgpu_buffer_rsrc_t cur_rsrc, next_rsrc;
    uint32_t lane_off;
    float ring[PF];
    // The descriptor's inputs go through readfirstlane: they ARE wave-uniform (kernel arguments), but once the compiler
    // cannot prove it -- a lambda capture, a struct copy in the caller is enough -- it wraps EVERY buffer load of the stream in
    // a waterfall loop (v_readfirstlane x 4, compare, s_and_saveexec ...: 2.5x the kernel time, seen in round 4).
    __device__ __forceinline__ static __amdgpu_buffer_rsrc_t rsrc_of(const float* blob, uint32_t chunks) {
        const uint64_t v = reinterpret_cast<uint64_t>(blob);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        float* p = reinterpret_cast<float*>(((uint64_t)hi << 32) | lo);
        return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)(__builtin_amdgcn_readfirstlane(chunks) * kChunkBytes), 0x00020000);
    }
    __device__ __forceinline__ void start(uint32_t lane, const float* first_blob, uint32_t first_chunks) {
        cur_rsrc = next_rsrc = rsrc_of(first_blob, first_chunks);
        lane_off = lane * 4u;
#pragma unroll
        for (int i = 0; i < PF; ++i) ring[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(cur_rsrc, lane_off, (uint32_t)(i * 256), 0));
    }
    __device__ __forceinline__ void begin_pass(const float* blob, uint32_t chunks, const float* next_blob, uint32_t next_chunks) {
        cur_rsrc = rsrc_of(blob, chunks);
        next_rsrc = rsrc_of(next_blob, next_chunks);
    }
    template <int I, int FRAGS>
    __device__ __forceinline__ float take() {
        static_assert(FRAGS % PF == 0 || I < FRAGS, "pass length must be a multiple of the ring depth");
        const float v = ring[I % PF];
        if constexpr (I + PF < FRAGS) ring[I % PF] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(cur_rsrc, lane_off, (uint32_t)((I + PF) * 256), 0));
        else ring[I % PF] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(next_rsrc, lane_off, (uint32_t)((I + PF - FRAGS) * 256), 0));
        return v;
    }
    // consume dummy fragments up to the next multiple of PF so the ring stays aligned for the next pass
    template <int FRAGS, int I = FRAGS>
    __device__ __forceinline__ void end_pass() {
        if constexpr (I % PF != 0) {
            // the dummy slot must still be refilled with the NEXT pass's fragment that belongs there
            constexpr int kPadded = (FRAGS + PF - 1) / PF * PF;
            ring[I % PF] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(next_rsrc, lane_off, (uint32_t)((I + PF - kPadded) * 256), 0));
            end_pass<FRAGS, I + 1>();
        }
    }
};

template <typename T> struct is_weight_ring { static constexpr bool value = false; };
template <int PF> struct is_weight_ring<WeightRing<PF>> { static constexpr bool value = true; };

// FRAGS: fragments in the (ring-padded) pass -- where the per-wave ring rolls over into the next blob.
// b(S) must be a plain register read: any vector-ALU work between two MFMAs of the same wave stalls the matrix
// pipe (see pipe_layer_from_tiles).
template <int NSTEPS, int MT, int F0, int FRAGS, int S = 0, typename Src, typename BOp>
__device__ __forceinline__ void pipe_steps(Src& wp, f32x16 (&acc)[MT], BOp&& b) {
    if constexpr (S < NSTEPS) {
        const float bv = b(S);
        [&]<int... T>(std::integer_sequence<int, T...>) {
            ((acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(wp.template take<F0 + S * MT + T, FRAGS>(), bv, acc[T], 0, 0, 0)), ...);
        }(std::make_integer_sequence<int, MT>{});
        if constexpr (is_weight_ring<Src>::value) __builtin_amdgcn_sched_barrier(0);
        pipe_steps<NSTEPS, MT, F0, FRAGS, S + 1>(wp, acc, b);
    }
}

// fragments a layer occupies in its pass
constexpr int lane_layer_frags(int steps, int mt, bool bias) { return (steps + (bias ? 1 : 0)) * mt; }
constexpr int tile_layer_frags(int kt, int mt, bool bias) { return (16 * kt + (bias ? 1 : 0)) * mt; }

template <int MT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[MT]) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
}
// the bias step: B = 1 in lane half 0 (whose A fragment carries the bias), 0 in half 1
template <int MT, int F0, int FRAGS, typename Src>
__device__ __forceinline__ void bias_step(Src& wp, uint32_t lane, f32x16 (&acc)[MT]) {
    pipe_steps<1, MT, F0, FRAGS>(wp, acc, [&](int) { return lane < 32 ? 1.0f : 0.0f; });
}

// layer whose input is per-lane-packed registers; F0 = index of its first fragment within the pass
template <int STEPS, int MT, int F0, int FRAGS, bool BIAS = true, typename Src>
__device__ __forceinline__ void pipe_layer_from_lanes(Src& wp, uint32_t lane, const float (&in)[STEPS], f32x16 (&acc)[MT]) {
    zero_acc<MT>(acc);
    if constexpr (BIAS) bias_step<MT, F0, FRAGS>(wp, lane, acc);
    pipe_steps<STEPS, MT, F0 + (BIAS ? MT : 0), FRAGS>(wp, acc, [&](int s) { return in[s]; });
}
// max(x, 0) as exactly one v_max_f32 (fmaxf() adds a canonicalising v_max x, x in front)
__device__ __forceinline__ float relu1(float x) {
    float y;
    asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
    return y;
}

// one reduction step: MT MFMAs (one per output tile) sharing the B operand `bv`; FI = index of the step's first fragment
template <int MT, int FI, int FRAGS, typename Src>
__device__ __forceinline__ void pipe_one_step(Src& wp, f32x16 (&acc)[MT], const float bv) {
    [&]<int... T>(std::integer_sequence<int, T...>) {
        ((acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(wp.template take<FI + T, FRAGS>(), bv, acc[T], 0, 0, 0)), ...);
    }(std::make_integer_sequence<int, MT>{});
}

// B operands R0 .. R1-1 of input tile K: accumulator read (+ ReLU)
template <int K, int R0, int R1, bool RELU, int KT>
__device__ __forceinline__ void stage_operands(const f32x16 (&in)[KT], float (&bq)[16]) {
#pragma unroll
    for (int r = R0; r < R1; ++r) bq[r] = RELU ? relu1(in[K][r]) : in[K][r];
}
// after which step of tile K clump c (of CLUMPS) of tile K+1's operands is staged
constexpr int clump_of_step(int clumps, int s) {
    const int per = 16 / clumps;
    return (s % per == (per - 1) / 2) ? s / per : -1;
}

// ---- ReLU in the LDS atomic unit --------------------------------------------------------------------------------------
// A wave's own vector-ALU instructions stall its matrix pipe (above), LDS instructions do not.  ds_max_f32 on a zero-filled
// slot IS ReLU, and its data operand may be an accumulator register: tile K+1's 16 operands go  acc --ds_max_f32--> LDS
// --ds_read--> VGPR, then the slot is zeroed again, all issued between tile K's MFMAs; no v_accvgpr_read, no v_max.
// `slot`: this lane's column of a [16][64] float area private to the wave (slot[r * 64]), zero on entry and on exit.
constexpr int kLdsStageFloats = 16 * 64;
template <int K, int KT>
__device__ __forceinline__ void lds_stage_put(const f32x16 (&in)[KT], float* slot) {
#pragma unroll
    for (int r = 0; r < 16; ++r) __hip_atomic_fetch_max(slot + r * 64, in[K][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_stage_get(float* slot, float (&bq)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) bq[r] = slot[r * 64];
#pragma unroll
    for (int r = 0; r < 16; ++r) slot[r * 64] = 0.0f;
}
template <int K, int S, int KT, int MT, int W0, int FRAGS, typename Src>
__device__ __forceinline__ void pipe_tile_step_lds(Src& wp, const f32x16 (&in)[KT], f32x16 (&acc)[MT], float (&bq)[2][16], float* slot) {
    pipe_one_step<MT, W0 + (K * 16 + S) * MT, FRAGS>(wp, acc, bq[K & 1][S]);
    if constexpr (K + 1 < KT && S == 1) lds_stage_put<(K + 1 < KT ? K + 1 : K)>(in, slot);
    if constexpr (K + 1 < KT && S == 4) lds_stage_get(slot, bq[(K + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
}

// step S of input tile K in the software-pipelined form: the step's MFMAs, then (at a clump point) part of tile K+1's operands
template <int K, int S, int KT, int MT, int W0, int FRAGS, bool RELU_IN, int CLUMPS, typename Src>
__device__ __forceinline__ void pipe_tile_step(Src& wp, const f32x16 (&in)[KT], f32x16 (&acc)[MT], float (&bq)[2][16]) {
    pipe_one_step<MT, W0 + (K * 16 + S) * MT, FRAGS>(wp, acc, bq[K & 1][S]);
    constexpr int c = clump_of_step(CLUMPS, S);
    if constexpr (K + 1 < KT && c >= 0)
        stage_operands<(K + 1 < KT ? K + 1 : K), c * (16 / CLUMPS), (c + 1) * (16 / CLUMPS), RELU_IN>(in, bq[(K + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
}
template <int K, int S, int KT, int W0, int FRAGS, bool RELU_IN, int CLUMPS, typename Src>
__device__ __forceinline__ void pipe_tile_step16(Src& wp, const f32x16 (&in)[KT], f32x16& a0, f32x16& a1, float (&bq)[2][16]) {
    if constexpr (S % 2 == 0) a0 = __builtin_amdgcn_mfma_f32_16x16x1f32(wp.template take<W0 + K * 16 + S, FRAGS>(), bq[K & 1][S], a0, 0, 0, 0);
    else a1 = __builtin_amdgcn_mfma_f32_16x16x1f32(wp.template take<W0 + K * 16 + S, FRAGS>(), bq[K & 1][S], a1, 0, 0, 0);
    constexpr int c = clump_of_step(CLUMPS, S);
    if constexpr (K + 1 < KT && c >= 0)
        stage_operands<(K + 1 < KT ? K + 1 : K), c * (16 / CLUMPS), (c + 1) * (16 / CLUMPS), RELU_IN>(in, bq[(K + 1) & 1]);
    if constexpr (is_weight_ring<Src>::value) __builtin_amdgcn_sched_barrier(0);
}

// layer whose input is KT accumulator tiles of the previous layer.
// A wave's own vector-ALU instructions do not hide under its own fp32 MFMAs (tools/probe/valu_overlap_probe.hip,
// env_pass_probe.hip): an accumulator read + ReLU "as the operand is fetched" cost ~30 cycles per step = 8 % of an
// environment pass.  CLUMPS == 0: the 16 operands of an input tile are produced as ONE cluster of vector-ALU instructions
// ahead of that tile's 16 x MT MFMAs (< 3 %).  CLUMPS > 0: software-pipelined -- tile K+1's operands are staged in CLUMPS
// small clumps between tile K's steps into a second register set, so that only the first tile's cluster of a layer sits
// in front of MFMAs that wait for it.
template <int KT, int MT, int F0, int FRAGS, bool RELU_IN = false, bool BIAS = true, int CLUMPS = 0, typename Src>
__device__ __forceinline__ void pipe_layer_from_tiles(Src& wp, uint32_t lane, const f32x16 (&in)[KT], f32x16 (&acc)[MT], float* lds_slot = nullptr) {
    zero_acc<MT>(acc);
    if constexpr (BIAS) bias_step<MT, F0, FRAGS>(wp, lane, acc);
    constexpr int W0 = F0 + (BIAS ? MT : 0);
    if constexpr (CLUMPS == -1) {
        // CLUMPS == -1: ReLU through the LDS atomic unit for tiles 1 .. KT-1 (the first tile's operands have nothing to hide under)
        static_assert(RELU_IN, "LDS staging is a ReLU");
        float bq[2][16];
        stage_operands<0, 0, 16, true>(in, bq[0]);
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... KS>(std::integer_sequence<int, KS...>) {
            (pipe_tile_step_lds<KS / 16, KS % 16, KT, MT, W0, FRAGS>(wp, in, acc, bq, lds_slot), ...);
        }(std::make_integer_sequence<int, 16 * KT>{});
    } else if constexpr (CLUMPS == 0) {
        [&]<int... K>(std::integer_sequence<int, K...>) {
            ([&] {
                float bq[16];
                stage_operands<K, 0, 16, RELU_IN>(in, bq);
                if constexpr (is_weight_ring<Src>::value) __builtin_amdgcn_sched_barrier(0);
                pipe_steps<16, MT, W0 + K * 16 * MT, FRAGS>(wp, acc, [&](int s) { return bq[s]; });
            }(), ...);
        }(std::make_integer_sequence<int, KT>{});
    } else {
        static_assert(16 % CLUMPS == 0, "CLUMPS must divide 16");
        float bq[2][16];
        stage_operands<0, 0, 16, RELU_IN>(in, bq[0]);
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... KS>(std::integer_sequence<int, KS...>) {
            (pipe_tile_step<KS / 16, KS % 16, KT, MT, W0, FRAGS, RELU_IN, CLUMPS>(wp, in, acc, bq), ...);
        }(std::make_integer_sequence<int, 16 * KT>{});
    }
}

// ---- layers with at most 16 outputs (pack_linear16) -----------------------------------------------------------------
// d: the v_mfma_f32_16x16x1_4B_f32 accumulator of the layer for one 32-sample group (see pack_linear16).  Two accumulators
// alternate (a 16x16 MFMA issues every 32 cycles but its result is ready after 40) and are added at the end.
template <int KT, int F0, int FRAGS, bool RELU_IN = true, bool BIAS = true, int CLUMPS = 0, typename Src>
__device__ __forceinline__ void pipe_layer16_from_tiles(Src& wp, uint32_t lane, const f32x16 (&in)[KT], f32x16& d) {
    f32x16 a0, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.0f; a1[r] = 0.0f; }
    if constexpr (BIAS) {     // A = bias[l & 15] in lanes 0-31 (blocks 0 and 1: every sample once), B = 1 there
        a1 = __builtin_amdgcn_mfma_f32_16x16x1f32(wp.template take<F0, FRAGS>(), lane < 32 ? 1.0f : 0.0f, a1, 0, 0, 0);
        if constexpr (is_weight_ring<Src>::value) __builtin_amdgcn_sched_barrier(0);
    }
    constexpr int W0 = F0 + (BIAS ? 1 : 0);
    constexpr int kClumps = CLUMPS ? CLUMPS : 1;
    float bq[2][16];
    stage_operands<0, 0, 16, RELU_IN>(in, bq[0]);
    __builtin_amdgcn_sched_barrier(0);
    [&]<int... KS>(std::integer_sequence<int, KS...>) {
        (pipe_tile_step16<KS / 16, KS % 16, KT, W0, FRAGS, RELU_IN, kClumps>(wp, in, a0, a1, bq), ...);
    }(std::make_integer_sequence<int, 16 * KT>{});
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r] = a0[r] + a1[r];
}
// lane (Q = l >> 4, j = l & 15): lo[q] = output row 4 Q + q of the group's sample j, hi[q] = of its sample 16 + j
__device__ __forceinline__ void fold16(const f32x16& d, float (&lo)[4], float (&hi)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { lo[q] = d[q] + d[8 + q]; hi[q] = d[4 + q] + d[12 + q]; }
}
// Quarter exchange across the wave (v_permlane16_swap; a quarter = 16 lanes): afterwards
//   a = [a.q0 | b.q0 | a.q2 | b.q2],  b = [a.q1 | b.q1 | a.q3 | b.q3]
__device__ __forceinline__ void swap_quarters(float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
// fold16 outputs of group A (samples of lanes 0-31) and group B (lanes 32-63)  ->  for every lane, rows 0 .. 4 NQ - 1 of its own
// sample: a 4 x 4 transpose between "which of (A lo, A hi, B lo, B hi)" and "which lane quarter", two swaps per stage
template <int NQ>
__device__ __forceinline__ void rows_to_lanes(const float (&alo)[4], const float (&ahi)[4], const float (&blo)[4], const float (&bhi)[4],
                                              float (&out)[4 * NQ]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float x0 = alo[q], x1 = ahi[q], x2 = blo[q], x3 = bhi[q];
        swap_halves(x0, x2);          // x0 = [A lo.q0, A lo.q1, B lo.q0, B lo.q1]   x2 = [A lo.q2, A lo.q3, B lo.q2, B lo.q3]
        swap_halves(x1, x3);          // x1 = [A hi.q0, A hi.q1, B hi.q0, B hi.q1]   x3 = [A hi.q2, ...]
        swap_quarters(x0, x1);        // x0 = [A lo.q0, A hi.q0, B lo.q0, B hi.q0]: every lane's own sample, rows 0-3;  x1: rows 4-7
        out[q] = x0;
        if constexpr (NQ > 1) out[4 + q] = x1;
        if constexpr (NQ > 2) {
            swap_quarters(x2, x3);    // x2: rows 8-11, x3: rows 12-15
            out[8 + q] = x2;
            if constexpr (NQ > 3) out[12 + q] = x3;
        }
    }
}

constexpr int pass_chunks(int frags) { return (frags + kChunkFrags - 1) / kChunkFrags; }

}  // namespace envidr
