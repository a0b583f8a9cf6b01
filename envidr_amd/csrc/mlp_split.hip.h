// Split-precision dense layers on the fp16 matrix cores of gfx950 (v_mfma_f32_32x32x16_f16, 16x the fp32 MFMA rate).
//
// NOT the default path and never the headline number: the fp32 kernels (mlp_mfma.hip.h) are.  This is the optional
// "split" shading mode (FusedOptions.env_precision = "f16x2"), reported separately by bench.py with its own error
// against the same goldens.
//
// Every fp32 operand is carried as two halves,  v = hi + lo  with  hi = fp16(v),  lo = fp16(v - hi)  (up to 22 significand
// bits: `lo` is a subnormal half for |v| < 2^-3 or so, i.e. an absolute resolution of 2^-25 there -- the fp16 MFMAs of
// gfx950 honour subnormal operands, tools/probe/mfma_f16_probe.hip checks that on the device), and a product is three
// MFMAs into ONE fp32 accumulator that starts at the bias:
//     acc = bias;   acc += a_hi * b_hi + a_hi * b_lo + a_lo * b_hi
// The dropped a_lo * b_lo term is 2^-22 relative; products of halves are exact in the fp32 accumulation.  Three 8-pass
// MFMAs (96 cycles) replace the eight 16-pass fp32 MFMAs (512 cycles) that cover the same 16 x 32 x 32 block.
// (Two MFMAs on the same accumulator are two apart in the stream: 36 instead of 33 cycles per MFMA,
// tools/probe/mfma_dep_probe.hip.)
//
// Layout conventions follow mlp_mfma.hip.h: weights are the A operand, activations the B operand, a 32 x 32 output tile
// of layer n is consumed as the B operand of layer n + 1 from the registers it was accumulated in:
//   A fragment (one 16-deep reduction step, one 32-row output tile) = 64 lanes x 8 halves (16 B per lane, 1 KiB);
//     lane (m = lane & 31, h = lane >> 5), slot i  <->  W[32 t + m][ k(step, h, i) ]
//   k(step, h, i):  lane order  16 step + 8 h + i               (layer input = per-item feature registers)
//                   tile order  32 (step / 2) + tile_row(8 (step & 1) + i, h)      (layer input = accumulator tiles)
//   (v_mfma_f32_32x32x16_f16 multiplies slot (h, i) of A with slot (h, i) of B whatever k the hardware calls it:
//    tools/probe/mfma_f16_probe.hip checks exactly this and the D register -> row map on the device.)
// Output tiles are accumulated kSplitGroup at a time (two accumulators each), so a layer's blob is ordered
//   [group of tiles][step][tile in group][hi fragment, lo fragment]
// and the four waves of a workgroup -- one per SIMD, each with its own 32 items -- consume it in lock step through a
// double-buffered LDS ring (SplitWeightPipe): 1 KiB per MFMA per wave from L2 would be 5x what L2 delivers.
#pragma once
#include "mlp_mfma.hip.h"

#include <cstring>

namespace envidr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr float kSplitMaxAct = 60000.0f;          // activations are clamped here before the fp16 split (fp16 max 65504)
constexpr int kSplitGroup = 2;                    // output tiles accumulated together (their fp16 conversion drains in the next group's MFMA gaps)

enum SplitOrder : int { kSplitLaneOrder = 0, kSplitTileOrder = 1 };
__host__ __device__ constexpr int split_k(SplitOrder o, int s, int h, int i) {
    return o == kSplitLaneOrder ? 16 * s + 8 * h + i : 32 * (s >> 1) + tile_row(8 * (s & 1) + i, h);
}
constexpr uint32_t split_steps_for(SplitOrder o, uint32_t k_in) { return o == kSplitLaneOrder ? (k_in + 15) / 16 : round_up(k_in, 32) / 16; }
constexpr int split_layer_frags(int steps, int mt) { return steps * mt * 2; }        // fragments of 1 KiB
constexpr int kSplitFragHalves = 64 * 8;

// ---- host side: fp32 -> (hi, lo) fp16 pairs, fragment order ---------------------------------------------------------
inline uint16_t f32_to_f16_rne(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u, ax = x & 0x7fffffffu;
    if (ax > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
    if (ax >= 0x47800000u) return (uint16_t)(sign | 0x7c00u);                     // >= 65536
    if (ax < 0x33000001u) return (uint16_t)sign;                                  // <= 2^-25: rounds to zero (tie to even)
    const int e = (int)(ax >> 23) - 127;
    const uint32_t m = (ax & 0x7fffffu) | 0x800000u;
    const int shift = e >= -14 ? 13 : -e - 1;
    uint32_t q = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) ++q;
    const uint32_t out = e >= -14 ? ((uint32_t)(e + 15) << 10) + (q - 0x400u) : q;   // a rounding carry walks into the exponent
    return (uint16_t)(sign | out);
}
inline float f16_bits_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) x = sign;
        else {
            int k = 0;
            uint32_t mm = m;
            while (!(mm & 0x400u)) { mm <<= 1; ++k; }
            x = sign | ((uint32_t)(127 - 15 - k + 1) << 23) | ((mm & 0x3ffu) << 13);
        }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}

// dst: split_layer_frags(steps, mt) * kSplitFragHalves halves.  W is [m_out, k_in] row-major.
inline void pack_split_weight(const float* W, uint32_t m_out, uint32_t k_in, SplitOrder order, uint32_t group, uint16_t* dst) {
    const uint32_t steps = split_steps_for(order, k_in), mt = round_up(m_out, 32) / 32;
    size_t frag = 0;
    for (uint32_t t0 = 0; t0 < mt; t0 += group)
        for (uint32_t s = 0; s < steps; ++s)
            for (uint32_t t = t0; t < std::min(mt, t0 + group); ++t, frag += 2)
                for (uint32_t lane = 0; lane < 64; ++lane)
                    for (uint32_t i = 0; i < 8; ++i) {
                        const uint32_t m = 32 * t + (lane & 31u), k = (uint32_t)split_k(order, (int)s, (int)(lane >> 5), (int)i);
                        const float w = (m < m_out && k < k_in) ? W[(size_t)m * k_in + k] : 0.0f;
                        const uint16_t hi = f32_to_f16_rne(w);
                        const uint16_t lo = f32_to_f16_rne(w - f16_bits_to_f32(hi));
                        dst[frag * kSplitFragHalves + lane * 8 + i] = hi;
                        dst[(frag + 1) * kSplitFragHalves + lane * 8 + i] = lo;
                    }
}

// ---- device side ------------------------------------------------------------------------------------------------------
constexpr int kSplitChunkFrags = 32;            // a chunk must outlast the L2 round trip of its successor's
constexpr uint32_t kSplitChunkBytes = kSplitChunkFrags * 1024u;       // prefetch: 32 KiB = 48 MFMAs = 1536 cycles; two of them in LDS
constexpr int kSplitStage = kSplitChunkFrags / 4;                     // 16-byte registers per lane holding a quarter chunk in flight
constexpr int split_pass_chunks(int frags) { return (frags + kSplitChunkFrags - 1) / kSplitChunkFrags; }

// WeightPipe (mlp_mfma.hip.h) for 1-KiB fragments, with one difference: the staged copy global -> registers -> LDS of the
// NEXT chunk is not done in one burst at the chunk boundary but one 1-KiB piece per wave every four fragments.  LDS takes
// writes at 64 B/clk: the 32 KiB of a chunk written at once kept it busy for ~500 cycles, and the fragment reads of the
// chunk just started queued behind them (the boundaries cost ~740 cycles each, a quarter of the kernel).
struct SplitWeightPipe {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4* lds;                 // workgroup base of u32x4[2][kSplitChunkFrags * 64]
    uint32_t lane, wave, slot;
    const u32x4* frag;          // this lane's column of the chunk being consumed
    u32x4 stage[kSplitStage];   // this wave's quarter of the next chunk (landed) / of the one after (in flight)
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t chunks;            // chunks in the (only) pass: the stream wraps around to chunk 0
    uint32_t local;
    uint32_t ahead_off;         // byte offset of the chunk whose loads are issued during the current chunk

    // wave w moves fragments [w * kSplitStage, (w + 1) * kSplitStage) of every chunk
    __device__ __forceinline__ uint32_t voff() const { return (wave * (kSplitStage * 64u) + lane) * 16u; }
    template <int I>
    __device__ __forceinline__ void load_piece(uint32_t chunk_off) {
        stage[I] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff(), chunk_off + (uint32_t)I * 1024u, 0);
    }
    template <int I>
    __device__ __forceinline__ void store_piece(uint32_t to_slot) {
        u32x4* dst = lds + to_slot * (kSplitChunkFrags * 64u) + wave * (kSplitStage * 64u) + lane;
        dst[I * 64] = stage[I];
    }
    __device__ __forceinline__ void start(void* lds_base, uint32_t lane_, uint32_t wave_, const void* blob, uint32_t chunks_) {
        lds = reinterpret_cast<u32x4*>(lds_base); lane = lane_; wave = wave_; chunks = chunks_;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(blob), 0, (int)(chunks_ * kSplitChunkBytes), 0x00020000);
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (load_piece<I>(0), ...);
            (store_piece<I>(1), ...);                          // chunk 0 -> slot 1 (the first boundary flips to it)
            (load_piece<I>(chunks_ > 1 ? kSplitChunkBytes : 0u), ...);        // chunk 1 staged
        }(std::make_integer_sequence<int, kSplitStage>{});
        slot = 0;
        local = 0xffffffffu;
        ahead_off = 0;
        frag = lds + lane;
    }
    __device__ __forceinline__ void boundary() {
        __syncthreads();                                       // everybody is done reading the other slot, and done writing this one
        ++local;
        if (local == chunks) local = 0;                        // the next pass streams the same blob again
        slot ^= 1u;
        uint32_t ahead = local + 2;
        if (ahead >= chunks) ahead -= chunks;
        if (ahead >= chunks) ahead -= chunks;
        ahead_off = ahead * kSplitChunkBytes;
        frag = lds + slot * (kSplitChunkFrags * 64u) + lane;
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int I>
    __device__ __forceinline__ half8 take() {
        constexpr int c = I % kSplitChunkFrags;
        if constexpr (c == 0) boundary();
        // piece c / 4 of the staged chunk (local + 1) goes to the other slot, and its register is refilled from chunk local + 2
        if constexpr (c % 4 == 1 && c / 4 < kSplitStage) {
            store_piece<c / 4>(slot ^ 1u);
            load_piece<c / 4>(ahead_off);
        }
        return __builtin_bit_cast(half8, frag[c * 64]);
    }
};

// Fragments are read from LDS PF ahead of the MFMAs that consume them, through a register ring (fragment I of a pass sits
// in ring[I % PF]; taking it re-issues the read of fragment I + PF, rolling over into the next pass at the end): left to
// the compiler, the four ds_read_b128 of a step are issued right in front of its MFMAs and every step waits out the LDS
// latency (measured: 80 k instead of 30 k cycles per pass).  FRAGS must be a multiple of PF (end_pass pads).
constexpr int kSplitAhead = 8;                    // fragments read from LDS ahead of the MFMAs that consume them
template <int PF>
struct SplitFragRing {
    SplitWeightPipe pipe;
    half8 ring[PF];
    __device__ __forceinline__ void start(void* lds_base, uint32_t lane, uint32_t wave, const void* blob, uint32_t chunks) {
        pipe.start(lds_base, lane, wave, blob, chunks);
        [&]<int... I>(std::integer_sequence<int, I...>) { ((ring[I] = pipe.template take<I>()), ...); }(std::make_integer_sequence<int, PF>{});
    }
    template <int I, int FRAGS>
    __device__ __forceinline__ half8 take() {
        static_assert(FRAGS % PF == 0, "pad the pass to a multiple of the ring depth");
        const half8 v = ring[I % PF];
        if constexpr (I + PF < FRAGS) ring[I % PF] = pipe.template take<I + PF>();
        else ring[I % PF] = pipe.template take<I + PF - FRAGS>();
        return v;
    }
    // dummy takes up to the padded length (keeps ring slot = index % PF valid in the next pass)
    template <int USED, int FRAGS, int I = USED>
    __device__ __forceinline__ void end_pass() {
        if constexpr (I < FRAGS) { (void)take<I, FRAGS>(); end_pass<USED, FRAGS, I + 1>(); }
    }
};

// v -> (hi, lo)
__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// accumulator registers 2 J, 2 J + 1 of a tile (bias included) -> ReLU -> their two slots of the reduction step (h, l) they belong to
template <int J>
__device__ __forceinline__ void split_pair_to_step(const f32x16& v, half8& h, half8& l) {
    constexpr int r = 2 * J;
    const float a = __builtin_amdgcn_fmed3f(v[r], 0.0f, kSplitMaxAct), b = __builtin_amdgcn_fmed3f(v[r + 1], 0.0f, kSplitMaxAct);
    _Float16 hi, lo;
    split_f16(a, hi, lo); h[r % 8] = hi; l[r % 8] = lo;
    split_f16(b, hi, lo); h[(r + 1) % 8] = hi; l[(r + 1) % 8] = lo;
}

// ---- the same layers with the conversion epilogue of a tile group slipped into the MFMA gaps of the NEXT group ------------
// A wave's own vector-ALU instructions hide under its MFMAs only if they sit between them in program order (one wave per
// SIMD: there is nobody else to fill the gaps), and the epilogue -- accumulator read, ReLU, fp16 split of every value: ~5
// instructions per activation, a third as many cycles as the MFMAs of the pass -- came after a group's last MFMA.  Here the
// accumulators of a finished group are *pending*: a Drain object converts them a few values at a time after each MFMA of the
// group that follows (two accumulator sets alternate), also across a layer boundary: the last group of layer n feeds only the
// LAST reduction steps of layer n + 1, so it is drained during the steps before those.
struct NoFiller {
    static constexpr int kDeadline = 0;
    template <int P, int NP> __device__ __forceinline__ void piece() const {}
};

// GT pending tiles, first of them tile T0 of its layer; sink(tile, j, acc) converts accumulator registers 2 j, 2 j + 1 of
// that tile; DEADLINE: the consuming group must be done draining before this reduction step (0 = any step will do)
template <int GT, int T0, int DEADLINE, class Sink>
struct Drain {
    static constexpr int kDeadline = DEADLINE;
    static constexpr int kPairs = GT * 8;
    const f32x16* acc;
    Sink* sink;
    template <int P, int NP>
    __device__ __forceinline__ void piece() const {
        constexpr int lo = kPairs * P / NP, hi = kPairs * (P + 1) / NP;
        [&]<int... Q>(std::integer_sequence<int, Q...>) {
            ((*sink)(std::integral_constant<int, T0 + (lo + Q) / 8>{}, std::integral_constant<int, (lo + Q) % 8>{}, acc[(lo + Q) / 8]), ...);
        }(std::make_integer_sequence<int, hi - lo>{});
    }
};

template <int S, int NSTEPS, int GT, int FG, int FRAGS, class Ring, class Filler>
__device__ __forceinline__ void split_steps_filled(Ring& wp, const half8 (&xh)[NSTEPS], const half8 (&xl)[NSTEPS], f32x16* acc, const Filler& fill) {
    if constexpr (S < NSTEPS) {
        half8 ah[GT], al[GT];
        [&]<int... T>(std::integer_sequence<int, T...>) {
            ((ah[T] = wp.template take<FG + (S * GT + T) * 2, FRAGS>(), al[T] = wp.template take<FG + (S * GT + T) * 2 + 1, FRAGS>()), ...);
        }(std::make_integer_sequence<int, GT>{});
        __builtin_amdgcn_sched_barrier(0);
        constexpr int GAPS = 3 * GT;
        constexpr int ACT = Filler::kDeadline == 0 ? NSTEPS : (Filler::kDeadline < NSTEPS ? Filler::kDeadline : NSTEPS);
        [&]<int... Q>(std::integer_sequence<int, Q...>) {
            ([&] {
                constexpr int kind = Q / GT, t = Q % GT;          // 0: a_hi b_hi, 1: a_hi b_lo, 2: a_lo b_hi
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kind == 2 ? al[t] : ah[t], kind == 1 ? xl[S] : xh[S], acc[t], 0, 0, 0);
                if constexpr (S < ACT) fill.template piece<S * GAPS + Q, ACT * GAPS>();
            }(), ...);
        }(std::make_integer_sequence<int, GAPS>{});
        __builtin_amdgcn_sched_barrier(0);          // one scheduling fence per reduction step
        split_steps_filled<S + 1, NSTEPS, GT, FG, FRAGS>(wp, xh, xl, acc, fill);
    }
}

// One layer, pipelined.  `cur` / `other`: the two accumulator sets (kSplitGroup tiles each); group 0 accumulates into `cur`
// while `first` (the pending work handed over by the caller: the previous layer's last group, or NoFiller) drains; group g + 1
// drains group g through `sink`.  Returns, through `done(acc, T0 of the last group, GT of it)`, the set that holds the
// layer's last group, still to be drained by whoever comes next.
template <int NSTEPS, int MT, int F0, int FRAGS, int T0 = 0, class Ring, class Bias, class Sink, class First, class Then>
__device__ __forceinline__ void split_layer_piped(Ring& wp, const half8 (&xh)[NSTEPS], const half8 (&xl)[NSTEPS], Bias&& bias, Sink& sink,
                                                  f32x16* cur, f32x16* other, const First& first, Then&& then) {
    constexpr int G = kSplitGroup, GT = (MT - T0) < G ? (MT - T0) : G, FG = F0 + 2 * NSTEPS * T0;
#pragma unroll
    for (int t = 0; t < GT; ++t) cur[t] = bias(T0 + t);
    split_steps_filled<0, NSTEPS, GT, FG, FRAGS>(wp, xh, xl, cur, first);
    if constexpr (T0 + G < MT) {
        const Drain<GT, T0, 0, Sink> pending{cur, &sink};
        split_layer_piped<NSTEPS, MT, F0, FRAGS, T0 + G>(wp, xh, xl, bias, sink, other, cur, pending, then);
    } else {
        then(cur, other, std::integral_constant<int, T0>{}, std::integral_constant<int, GT>{});
    }
}

}  // namespace envidr
