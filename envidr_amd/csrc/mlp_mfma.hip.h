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
    __device__ __forceinline__ static __amdgpu_buffer_rsrc_t rsrc_of(const float* blob, uint32_t chunks) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(blob), 0, (int)(chunks * kChunkBytes), 0x00020000);
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

// layer whose input is KT accumulator tiles of the previous layer.
// The 16 B operands of one input tile are produced (accumulator read + ReLU) as ONE cluster of vector-ALU
// instructions ahead of that tile's 16 x MT MFMAs.  Measured on gfx950 (tools/probe/valu_overlap_probe.hip,
// env_pass_probe.hip): a wave's own vector-ALU instruction between two of its MFMAs costs ~10 cycles of matrix-pipe
// idle time plus ~20 more when the MFMA consumes its result, so ReLU "as the operand is fetched" (one small VALU
// chain per step) cost 30 cycles per step = 8 % of an environment pass; one cluster per tile costs < 1 %.
template <int KT, int MT, int F0, int FRAGS, bool RELU_IN = false, bool BIAS = true, typename Src>
__device__ __forceinline__ void pipe_layer_from_tiles(Src& wp, uint32_t lane, const f32x16 (&in)[KT], f32x16 (&acc)[MT]) {
    zero_acc<MT>(acc);
    if constexpr (BIAS) bias_step<MT, F0, FRAGS>(wp, lane, acc);
    [&]<int... K>(std::integer_sequence<int, K...>) {
        ([&] {
            float bq[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bq[r] = RELU_IN ? relu1(in[K][r]) : in[K][r];
            if constexpr (is_weight_ring<Src>::value) __builtin_amdgcn_sched_barrier(0);
            pipe_steps<16, MT, F0 + (BIAS ? MT : 0) + K * 16 * MT, FRAGS>(wp, acc, [&](int s) { return bq[s]; });
        }(), ...);
    }(std::make_integer_sequence<int, KT>{});
}

constexpr int pass_chunks(int frags) { return (frags + kChunkFrags - 1) / kChunkFrags; }

}  // namespace envidr
