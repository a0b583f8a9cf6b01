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
constexpr uint32_t kSplitBlobGrain = 32u * 1024u; // the host pads a blob to a multiple of this (zero fragments)
constexpr int kSplitChunkFrags = 8;             // a chunk of the streamed part: 8 KiB = 12 MFMAs = ~0.45 k cycles
constexpr int kSplitSlots = 8;                  // chunk slots in LDS (64 KiB)
constexpr int kSplitPieces = kSplitChunkFrags / 4;                    // 1-KiB pieces of a chunk each of the four waves moves
constexpr int kSplitAhead = 8;                  // fragments read from LDS ahead of the MFMAs that consume them
constexpr int kSplitMeetAt = 4;                 // read index inside a chunk at which the block meets (see SplitWeightPipe)
constexpr int kSplitResident = 88;              // fragments at the head of a pass that stay in LDS for the whole kernel
constexpr uint32_t kSplitLdsBytes = (uint32_t)(kSplitSlots * kSplitChunkFrags + kSplitResident) * 1024u;

// The weight stream of a workgroup.  Four waves in lock step read the same 1-KiB fragments from LDS; what bounds the kernel is
// the bandwidth at which the 256 CUs can pull that stream out of L2 (every CU streams the same 624 KiB per 32 x 4 items: 6.2 TB/s
// measured, the chip's L2 -> LDS fill ceiling), so
//   * the first kSplitResident fragments of a pass are loaded ONCE per workgroup and stay resident (LDS is 160 KiB);
//   * the rest is streamed through a ring of eight 8-KiB chunk slots that the waves fill, a quarter each, with LDS-DMA --
//     `buffer_load_dwordx4 ... lds`: global -> LDS without a register or a ds_write in between, wave-uniform destination
//     M0 + 16 B per lane, which IS the fragment layout.
//
// One s_barrier per chunk, at read index kSplitMeetAt of chunk b ("B_b"), with no LDS wait in front of it:
//   * the register ring is kSplitAhead = 8 fragments deep and a reduction step takes four, so at B_b every read of chunk b-2
//     has been consumed by an MFMA that precedes B_b in program order: after B_b the slot of chunk b-2 is overwritten with
//     chunk b+6 (two pieces per wave, issued in the fragments that follow);
//   * before B_{b+1} a wave waits for all but its eight youngest vector-memory operations (`vmcnt(8)`: the pieces of chunks
//     b+3 .. b+6) -- so its pieces of chunk b+2, issued after B_{b-4}, have landed; after the barrier that holds for all four
//     waves, and chunk b+2 is read from four fragments later on.  A piece has five chunk periods (~2 k cycles) to land.  (Other
//     vector-memory operations of the wave -- the round's own loads and stores -- are younger than those pieces and only make
//     the wait stricter; the counter retires in issue order.)
// The compiler does not know about the DMA (inline asm): it neither counts it nor orders LDS reads behind it; the two rules
// above are what orders them.
struct SplitWeightPipe {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const u32x4* res;           // this lane's column of the resident fragments
    const u32x4* ring;          // this lane's column of slot 0
    const u32x4* frag;          // this lane's column of the chunk being read
    u32x4 rsrc;                 // raw buffer descriptor of the blob
    uint32_t voff;              // this lane's byte offset inside a chunk: wave * 2 KiB + lane * 16
    uint32_t lds_wave;          // LDS byte address of this wave's first piece in slot 0
    uint32_t chunks;            // chunks in the streamed part of the (only) pass: the stream wraps around to chunk 0
    uint32_t slot;              // slot of the chunk being read
    uint32_t fill_off, fill_lds;    // blob byte offset / LDS byte address (this wave's pieces) of the chunk being filled
    uint32_t next_fill;         // chunk the next barrier starts to fill

    __device__ __forceinline__ void dma(uint32_t lds_dst, uint32_t v, uint32_t blob_off) const {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(lds_dst), "v"(v), "s"(rsrc), "s"(blob_off)
                     : "memory");
    }
    template <int P>
    __device__ __forceinline__ void dma_piece() const { dma(fill_lds + (uint32_t)P * 1024u, voff, fill_off + (uint32_t)P * 1024u); }
    __device__ __forceinline__ void aim(uint32_t chunk, uint32_t to_slot) {
        fill_off = (uint32_t)kSplitResident * 1024u + chunk * (uint32_t)(kSplitChunkFrags * 1024);
        fill_lds = lds_wave + to_slot * (uint32_t)(kSplitChunkFrags * 1024);
    }
    // `frags`: fragments of a pass (a multiple of kSplitAhead, and of kSplitChunkFrags past the resident part)
    __device__ __forceinline__ void start(void* lds_base, uint32_t lane, uint32_t wave, const void* blob, uint32_t frags) {
        static_assert(kSplitSlots == 8 && kSplitChunkFrags == 8 && kSplitAhead == 8 && kSplitMeetAt == 4 && kSplitResident % 4 == 0, "the schedule in the comment above");
        const u32x4* lds = reinterpret_cast<const u32x4*>(lds_base);
        ring = lds + lane;
        res = lds + kSplitSlots * kSplitChunkFrags * 64 + lane;
        chunks = (frags - (uint32_t)kSplitResident) / (uint32_t)kSplitChunkFrags;
        const uint64_t addr = (uint64_t)blob;
        rsrc = u32x4{(uint32_t)addr, (uint32_t)(addr >> 32) & 0xffffu, (frags * 1024u + kSplitBlobGrain - 1u) / kSplitBlobGrain * kSplitBlobGrain, 0x00020000u};
#pragma unroll
        for (int q = 0; q < 4; ++q) rsrc[q] = __builtin_amdgcn_readfirstlane(rsrc[q]);
        const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)lds_base);
        voff = (wave * (uint32_t)kSplitPieces * 64u + lane) * 16u;
        lds_wave = lds0 + wave * (uint32_t)kSplitPieces * 1024u;
        // resident fragments: wave w moves fragments [w, w + 1) * kSplitResident / 4
        constexpr uint32_t per_wave = kSplitResident / 4;
        for (uint32_t q = 0; q < per_wave; ++q)
            dma(lds0 + (uint32_t)(kSplitSlots * kSplitChunkFrags) * 1024u + (wave * per_wave + q) * 1024u, lane * 16u, (wave * per_wave + q) * 1024u);
        // chunks 0 .. 5 -> slots 0 .. 5 (the first barrier starts chunk 6)
        uint32_t c = 0;
        for (uint32_t sl = 0; sl < (uint32_t)kSplitSlots - 2u; ++sl) {
            aim(c, sl);
            [&]<int... P>(std::integer_sequence<int, P...>) { (dma_piece<P>(), ...); }(std::make_integer_sequence<int, kSplitPieces>{});
            if (++c == chunks) c = 0;
        }
        next_fill = c;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        slot = kSplitSlots - 1;            // the first streamed take() steps to slot 0
        frag = ring;
    }
    __device__ __forceinline__ void next_chunk() {
        slot = (slot + 1u) & (uint32_t)(kSplitSlots - 1);
        frag = ring + slot * (uint32_t)(kSplitChunkFrags * 64);
    }
    __device__ __forceinline__ void meet() {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kSplitPieces * (kSplitSlots - 4)) : "memory");
        aim(next_fill, (slot + (uint32_t)kSplitSlots - 2u) & (uint32_t)(kSplitSlots - 1));          // the slot of chunk b - 2
        if (++next_fill == chunks) next_fill = 0;
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int I>
    __device__ __forceinline__ half8 take() {
        if constexpr (I < kSplitResident) return __builtin_bit_cast(half8, res[I * 64]);
        else {
            constexpr int c = (I - kSplitResident) % kSplitChunkFrags;
            if constexpr (c == 0) next_chunk();
            if constexpr (c == kSplitMeetAt) meet();
            if constexpr (c == kSplitMeetAt + 1) dma_piece<0>();
            if constexpr (c == kSplitMeetAt + 3) dma_piece<1>();
            return __builtin_bit_cast(half8, frag[c * 64]);
        }
    }
};

// Fragments are read from LDS PF ahead of the MFMAs that consume them, through a register ring (fragment I of a pass sits
// in ring[I % PF]; taking it re-issues the read of fragment I + PF, rolling over into the next pass at the end): left to
// the compiler, the four ds_read_b128 of a step are issued right in front of its MFMAs and every step waits out the LDS
// latency (measured: 80 k instead of 30 k cycles per pass).  FRAGS must be a multiple of PF (end_pass pads).
template <int PF>
struct SplitFragRing {
    SplitWeightPipe pipe;
    half8 ring[PF];
    __device__ __forceinline__ void start(void* lds_base, uint32_t lane, uint32_t wave, const void* blob, uint32_t frags) {
        pipe.start(lds_base, lane, wave, blob, frags);
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
