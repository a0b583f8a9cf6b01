// Split-precision environment MLP, second form: layers fused in pairs, two waves per SIMD, one weight stream for eight waves.
//
// mlp_split.hip.h keeps a group's (hi, lo) activations -- layer input + layer output, 256 registers -- in a 512-register wave, one wave per
// SIMD: the wave's own conversion / LDS / DMA instructions sit between its MFMAs (57 % matrix-pipe busy), its accumulators
// live in AGPRs (every vector-ALU access to one is a v_accvgpr move), and each 1-KiB weight fragment pulled through L2 -> LDS serves 128 items
// per workgroup.  Here a layer's OUTPUT buffer does not exist:
//
//     phase A:  for each 32-row tile t of layer 1:   y1_t = relu(W1[t] in + b1[t])          (its own accumulator, 16 registers)
//                                                     acc2[u] += W2[u][:, tile t] y1_t  for every tile u of layer 2
//     in place: x2 = (hi, lo)(relu(acc2))                                                    (same registers: 32 bits per value either way)
//     phase B:  for each tile t of layer 3:          y3_t = relu(W3[t] x2 + b3[t])
//                                                     acc4 += W4[:, tile t] y3_t
//
// so a wave's 32 items need ONE 256-wide buffer (acc2, then x2: 128 registers) + a tile in flight, the kernel fits 256 registers, two waves
// share a SIMD (one wave's conversions and LDS reads issue under the other's MFMAs; no AGPRs, MFMA results are ordinary VGPRs), and the eight
// waves of a workgroup -- 256 items -- consume the weight stream in lock step: half the L2 -> LDS traffic per item.  The first layer's input
// (2 x TERMS IDE features per item as (hi, lo) fragments) waits in LDS, written there by the round's prologue, and is read as the B operand
// of layer 1 once per output tile.
//
// Per-accumulator order of additions is the one of mlp_split.hip.h (bias first; per 16-deep step hi*hi, hi*lo, lo*hi; steps in
// ascending order), and the same fp32 -> (hi, lo) split: the two kernels produce the same bits (tests/test_split_gpu.py).
//
// Weight stream: no resident part (LDS holds the layer-1 operands instead); a ring of eight 8-KiB chunk slots filled by LDS-DMA, one
// 1-KiB piece per wave per chunk; one bare s_barrier per chunk with a counted vmcnt in front -- the protocol of SplitWeightPipe with these
// numbers (see there for the ordering argument):
//   * at barrier B_b (read index 4 of chunk b) every wave has CONSUMED all fragments below 8 b (the register ring is four deep), i.e. chunk
//     b - 1: its slot is aimed at chunk b + 7 there, and that piece is ISSUED when the wave arrives at B_{b+1}, in front of the wait -- the
//     waves reach a barrier at different times, so their eight DMA instructions pass through the CU's one address path while the early ones
//     wait anyway (issued right behind a barrier they queue up there with nobody issuing an MFMA: 1.6 of 11.3 ms);
//   * in front of B_j a wave waits for all but its five youngest vector-memory operations (the pieces of chunks j + 2 .. j + 6): its piece
//     of chunk j + 1 has landed; after the barrier that holds for all eight waves, and chunk j + 1 is read from four fragments later on.  A
//     piece has five chunk periods to land.
#pragma once
#include "mlp_split.hip.h"

namespace envidr {

constexpr int kS2Waves = 8;                    // waves per workgroup: two per SIMD
constexpr int kS2ChunkFrags = 8;               // fragments per chunk: one 1-KiB piece per wave
constexpr int kS2Slots = 8;
constexpr int kS2Ahead = 4;                    // register ring: fragments read from LDS ahead of their MFMAs (two (hi, lo) pairs = 6 MFMAs)
constexpr int kS2MeetAt = 4;
constexpr uint32_t kS2RingBytes = (uint32_t)(kS2Slots * kS2ChunkFrags) * 1024u;

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}

// Fragment order of a pass = the order the kernel consumes them in (kernel and host packer agree on this and on nothing else).  The
// finished tile t of layer 1 (3) is converted to (hi, lo) in the MFMA gaps of a block that does not depend on it, so tile t + 1 is computed
// before tile t is consumed:
//     phase A:  A1(0) A1(1) | A2(0) A1(2) | A2(1) A1(3) | ... | A2(T-2) | A2(T-1)        A1(t): layer-1 tile t, S1 steps
//                                                                                         A2(t): steps 2t, 2t+1 of every layer-2 tile
//     phase B:  B3(0) | B3(1) B4(0) | B3(2) B4(1) | ... | B4(T-1)                          B3(t): layer-3 tile t, 2 T steps;  B4(t): last layer
template <int TERMS, int ENV_T>
struct Split2Layout {
    static constexpr int K1 = 2 * TERMS, S1 = (K1 + 15) / 16, SH = 2 * ENV_T;
    static constexpr int A1 = 2 * S1, A2 = 4 * ENV_T;
    static constexpr int B3 = 2 * SH, B4 = 4;
    static constexpr int FB = ENV_T * (A1 + A2), Frags = FB + ENV_T * (B3 + B4);
    static constexpr int Padded = (Frags + kS2ChunkFrags - 1) / kS2ChunkFrags * kS2ChunkFrags;
    static_assert(kS2ChunkFrags % kS2Ahead == 0, "one padding serves the chunk size and the register ring");
    static_assert(Padded >= kS2Slots * kS2ChunkFrags, "a pass shorter than the ring");
    static constexpr int BiasTiles = 3 * ENV_T + 1;
    static constexpr int InFrags = S1 * 2;                                       // layer-1 operands of a wave in LDS: [step][hi, lo]
    static constexpr int imin(int a, int b) { return a < b ? a : b; }
    static constexpr int imax(int a, int b) { return a > b ? a : b; }
    // blocks in front of A2(t): A1(0 .. min(t + 1, T - 1)) and A2(0 .. t - 1)
    static constexpr int a2_block(int t) { return imin(t + 2, ENV_T) * A1 + t * A2; }
    static constexpr int a1_block(int t) { return t < 2 ? t * A1 : a2_block(t - 2) + A2; }
    static constexpr int b4_block(int t) { return FB + imin(t + 2, ENV_T) * B3 + t * B4; }
    static constexpr int b3_block(int t) { return t == 0 ? FB : FB + t * B3 + (t - 1) * B4; }
    __host__ __device__ static constexpr int a1(int t, int s, int hl) { return a1_block(t) + 2 * s + hl; }
    __host__ __device__ static constexpr int a2(int t, int s, int u, int hl) { return a2_block(t) + s * 2 * ENV_T + 2 * u + hl; }
    __host__ __device__ static constexpr int b3(int t, int s, int hl) { return b3_block(t) + 2 * s + hl; }
    __host__ __device__ static constexpr int b4(int t, int s, int hl) { return b4_block(t) + 2 * s + hl; }
};

// ---- host side -------------------------------------------------------------------------------------------------------------
// Layer 1 in lane order (slot (step s, half h, i) <-> column 16 s + 8 h + i), layers 2-4 in tile order: split_k() of mlp_split.hip.h.
inline void split2_put(uint16_t* dst, int frag, uint32_t lane, uint32_t i, float w) {
    const uint16_t hi = f32_to_f16_rne(w);
    const uint16_t lo = f32_to_f16_rne(w - f16_bits_to_f32(hi));
    dst[(size_t)frag * kSplitFragHalves + lane * 8 + i] = hi;
    dst[(size_t)(frag + 1) * kSplitFragHalves + lane * 8 + i] = lo;
}
// W1 [H, 2 TERMS], W2, W3 [H, H], W4 [12, H] row-major, H = 32 ENV_T.  dst: Padded * kSplitFragHalves halves.
template <int TERMS, int ENV_T>
inline void pack_env_split2(const float* W1, const float* W2, const float* W3, const float* W4, uint16_t* dst) {
    using L = Split2Layout<TERMS, ENV_T>;
    constexpr uint32_t H = 32 * ENV_T;
    memset(dst, 0, sizeof(uint16_t) * (size_t)L::Padded * kSplitFragHalves);
    for (int t = 0; t < ENV_T; ++t) {
        for (int s = 0; s < L::S1; ++s)
            for (uint32_t lane = 0; lane < 64; ++lane)
                for (uint32_t i = 0; i < 8; ++i) {
                    const uint32_t m = 32 * t + (lane & 31u), k = (uint32_t)split_k(kSplitLaneOrder, s, (int)(lane >> 5), (int)i);
                    const float w = k < (uint32_t)L::K1 ? W1[(size_t)m * L::K1 + k] : 0.0f;
                    split2_put(dst, L::a1(t, s, 0), lane, i, w);
                }
        for (int s = 0; s < 2; ++s)
            for (int u = 0; u < ENV_T; ++u)
                for (uint32_t lane = 0; lane < 64; ++lane)
                    for (uint32_t i = 0; i < 8; ++i) {
                        const uint32_t m = 32 * u + (lane & 31u), k = (uint32_t)split_k(kSplitTileOrder, 2 * t + s, (int)(lane >> 5), (int)i);
                        split2_put(dst, L::a2(t, s, u, 0), lane, i, W2[(size_t)m * H + k]);
                    }
        for (int s = 0; s < L::SH; ++s)
            for (uint32_t lane = 0; lane < 64; ++lane)
                for (uint32_t i = 0; i < 8; ++i) {
                    const uint32_t m = 32 * t + (lane & 31u), k = (uint32_t)split_k(kSplitTileOrder, s, (int)(lane >> 5), (int)i);
                    split2_put(dst, L::b3(t, s, 0), lane, i, W3[(size_t)m * H + k]);
                }
        for (int s = 0; s < 2; ++s)
            for (uint32_t lane = 0; lane < 64; ++lane)
                for (uint32_t i = 0; i < 8; ++i) {
                    const uint32_t m = lane & 31u, k = (uint32_t)split_k(kSplitTileOrder, 2 * t + s, (int)(lane >> 5), (int)i);
                    split2_put(dst, L::b4(t, s, 0), lane, i, m < 12 ? W4[(size_t)m * H + k] : 0.0f);
                }
    }
}

// ---- device side -----------------------------------------------------------------------------------------------------------
struct Split2Stream {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const u32x4* ring;          // this lane's column of slot 0
    const u32x4* frag;          // this lane's column of the chunk being read
    u32x4 rsrc;
    uint32_t voff;              // this lane's byte offset inside a chunk: wave * 1 KiB + lane * 16
    uint32_t lds_wave;          // LDS byte address of this wave's piece in slot 0
    uint32_t chunks, slot, fill_off, fill_lds, next_fill;

    __device__ __forceinline__ void dma(uint32_t lds_dst, uint32_t blob_off) const {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(blob_off)
                     : "memory");
    }
    __device__ __forceinline__ void aim(uint32_t chunk, uint32_t to_slot) {
        fill_off = chunk * (uint32_t)(kS2ChunkFrags * 1024);
        fill_lds = lds_wave + to_slot * (uint32_t)(kS2ChunkFrags * 1024);
    }
    __device__ __forceinline__ void start(void* lds_base, uint32_t lane, uint32_t wave, const void* blob, uint32_t frags) {
        static_assert(kS2Waves == 8 && kS2Slots == 8 && kS2ChunkFrags == 8 && kS2Ahead == 4 && kS2MeetAt == 4, "the schedule in the header comment");
        ring = reinterpret_cast<const u32x4*>(lds_base) + lane;
        chunks = frags / (uint32_t)kS2ChunkFrags;
        const uint64_t addr = (uint64_t)blob;
        rsrc = u32x4{(uint32_t)addr, (uint32_t)(addr >> 32) & 0xffffu, frags * 1024u, 0x00020000u};
#pragma unroll
        for (int q = 0; q < 4; ++q) rsrc[q] = __builtin_amdgcn_readfirstlane(rsrc[q]);
        const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)lds_base);
        voff = (wave * 64u + lane) * 16u;
        lds_wave = lds0 + wave * 1024u;
        uint32_t c = 0;
        for (uint32_t sl = 0; sl < (uint32_t)kS2Slots - 2u; ++sl) {          // chunks 0 .. 5 -> slots 0 .. 5
            aim(c, sl);
            dma(fill_lds, fill_off);
            if (++c == chunks) c = 0;
        }
        aim(c, (uint32_t)kS2Slots - 2u);                                       // chunk 6 -> slot 6: issued at the first barrier
        if (++c == chunks) c = 0;
        next_fill = c;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        slot = 0;
        frag = ring;
    }
    __device__ __forceinline__ void next_chunk() {
        slot = (slot + 1u) & (uint32_t)(kS2Slots - 1);
        frag = ring + slot * (uint32_t)(kS2ChunkFrags * 64);
    }
    __device__ __forceinline__ void meet() {
        dma(fill_lds, fill_off);                          // the piece aimed at the last barrier
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kS2Slots - 3) : "memory");
        aim(next_fill, (slot + (uint32_t)kS2Slots - 1u) & (uint32_t)(kS2Slots - 1));          // the slot of chunk b - 1
        if (++next_fill == chunks) next_fill = 0;
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int I>
    __device__ __forceinline__ half8 take() {
        constexpr int c = I % kS2ChunkFrags;
        if constexpr (c == 0) next_chunk();
        if constexpr (c == kS2MeetAt) meet();
        return __builtin_bit_cast(half8, frag[c * 64]);
    }
    // the first kS2Ahead reads of the kernel (chunk 0, in front of the first meeting point)
    template <int I>
    __device__ __forceinline__ half8 first() const { return __builtin_bit_cast(half8, frag[I * 64]); }
};

template <int PF>
struct Split2Ring {
    Split2Stream pipe;
    half8 ring[PF];
    __device__ __forceinline__ void start(void* lds_base, uint32_t lane, uint32_t wave, const void* blob, uint32_t frags) {
        pipe.start(lds_base, lane, wave, blob, frags);
        static_assert(PF <= kS2MeetAt, "the start-up reads end in front of the first meeting point");
        static_for<PF>([&](auto i) { ring[i] = pipe.template first<decltype(i)::value>(); });
    }
    template <int I, int FRAGS>
    __device__ __forceinline__ half8 take() {
        static_assert(FRAGS % PF == 0, "pad the pass to a multiple of the ring depth");
        const half8 v = ring[I % PF];
        if constexpr (I + PF < FRAGS) ring[I % PF] = pipe.template take<I + PF>();
        else ring[I % PF] = pipe.template take<I + PF - FRAGS>();
        return v;
    }
    template <int USED, int FRAGS, int I = USED>
    __device__ __forceinline__ void end_pass() {
        if constexpr (I < FRAGS) { (void)take<I, FRAGS>(); end_pass<USED, FRAGS, I + 1>(); }
    }
};

struct NoFill {
    template <int Q, int NQ> __device__ __forceinline__ void piece() const {}
};

// pairs of accumulator registers of NT (group, tile) accumulators -> (hi, lo) halves, spread over the NQ MFMA gaps of the block that
// runs meanwhile: gap Q converts pairs [8 NT Q / NQ, 8 NT (Q + 1) / NQ).  sink(n, j, acc): pair j (registers 2 j, 2 j + 1) of accumulator n.
template <int NT, class Sink>
struct CvtFill {
    const Sink* sink;
    template <int Q, int NQ>
    __device__ __forceinline__ void piece() const {
        constexpr int lo = 8 * NT * Q / NQ, hi = 8 * NT * (Q + 1) / NQ;
        static_for<hi - lo>([&](auto p) { (*sink)(std::integral_constant<int, (lo + decltype(p)::value) / 8>{}, std::integral_constant<int, (lo + decltype(p)::value) % 8>{}); });
    }
};

// One 16-deep reduction step of GT output tiles: fragments I0 .. I0 + 2 GT - 1 of the pass ((hi, lo) per tile), the B operand (bh, bl),
// accumulators acc[t].  MFMA q of the step is gap Q0 + q of the running filler.
template <int GT, int I0, int FRAGS, int Q0, int NQ, class Ring, class Fill>
__device__ __forceinline__ void split2_step(Ring& wp, const half8& bh, const half8& bl, f32x16* acc, const Fill& fill) {
    half8 ah[GT], al[GT];
    static_for<GT>([&](auto t) {
        ah[t] = wp.template take<I0 + 2 * decltype(t)::value, FRAGS>();
        al[t] = wp.template take<I0 + 2 * decltype(t)::value + 1, FRAGS>();
    });
    __builtin_amdgcn_sched_barrier(0);
    static_for<3 * GT>([&](auto qc) {
        constexpr int q = decltype(qc)::value, kind = q / GT, t = q % GT;          // 0: a_hi b_hi, 1: a_hi b_lo, 2: a_lo b_hi
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kind == 2 ? al[t] : ah[t], kind == 1 ? bl : bh, acc[t], 0, 0, 0);
        fill.template piece<Q0 + q, NQ>();
    });
    __builtin_amdgcn_sched_barrier(0);
}

}  // namespace envidr
