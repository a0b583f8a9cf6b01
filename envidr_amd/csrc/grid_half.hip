// Half-precision instantiations of the feature-grid operators for gfx950: the `scalar_t = at::Half` side of the reference's
// AT_DISPATCH_FLOATING_TYPES_AND_HALF (hashencoder/src/hashencoder.cu:747,778; gridencoder/src/gridencoder.cu:443,474) --
// what `hashencoder/hashgrid.py:19` (`custom_fwd(cast_inputs=torch.half)`) and `gridencoder/grid.py:37-40` (half table under
// autocast) reach.  Table, outputs, dy_dx and gradients are fp16; hashencoder also takes fp16 inputs, gridencoder fp32.
//
// Arithmetic follows c10::Half (c10/util/Half-inl.h): every operation is done in float and a value is narrowed -- round to
// nearest even, `v_cvt_f16_f32` -- exactly where the reference's expressions make it an at::Half again:
//     results[ch] += w * grid[i]                 a = H(F(a) + F(H(w * F(g))))       (Half += float narrows the float first)
//     grid[right] - grid[left]                   H(F(r) - F(l))
//     grad * dy_dx, result += ...                H(F(g) * F(d)),  H(F(acc) + F(prod))
//     (__half)(w * grad_cur[c]) + atomic         H(w * F(g)), then an fp16 atomic add
// so the forward pass and the input gradient are bit-identical to the oracle's restatement (oracle/c, grid_point_level_h);
// the table gradient is a sum of fp16 atomic adds whose rounding depends on their order, as it does in the reference.
// One lane per (point, level); these instantiations exist for interface completeness, the fp32 kernels are the tuned ones.
#include "grid_core.hip.h"

// (the opaque statement in H() keeps the largest instantiation -- D = 5, C = 8: 32 corners x 8 channels -- from being fully unrolled;
//  that is a missed optimisation in a kernel that exists for interface completeness, not something to be warned about on every build)
#pragma clang diagnostic ignored "-Wpass-failed"

using namespace envidr;

namespace {

typedef _Float16 h16;
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float F(h16 v) { return (float)v; }
// narrow a float that EXISTS as a float: c10::Half narrows the fp32 result of `w * grid[i]` (two roundings).  Left alone, the compiler
// folds `(h16)(a * b)` into v_fma_mixlo_f16 -- the exact product rounded ONCE to fp16 -- which differs from the reference where the
// fp32 rounding moves the product across an fp16 rounding boundary (seen on ~1 value in 2 000, tools/fuzz_ops.py); the empty asm makes
// the fp32 value materialise first.  (Sums and differences of two halves are immune: fp32 holds them with 2 * 11 + 2 bits to spare.)
__device__ __forceinline__ h16 H(float v) {
    asm("" : "+v"(v));          // (not volatile: it only has to be opaque, and a volatile statement keeps the big instantiations from unrolling)
    return (h16)v;
}
__device__ __forceinline__ h16 add_f(h16 a, float b) { return H(F(a) + F(H(b))); }       // Half += float

template <typename IN> __device__ __forceinline__ float widen(IN v) { return (float)v; }

// kernel_grid<at::Half, D, C> (hashencoder.cu:103-254 with SMOOTH, gridencoder.cu:75-223 without)
template <int D, int C, bool SMOOTH, typename IN>
__global__ void __launch_bounds__(kBlock) k_grid_forward_h(const IN* __restrict__ inputs, const h16* __restrict__ embeddings,
                                                           const int32_t* __restrict__ offsets, h16* __restrict__ outputs, uint32_t B,
                                                           uint32_t L, LevelScale ls, h16* __restrict__ dy_dx, uint32_t gridtype,
                                                           bool align_corners) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t level = blockIdx.y;
    if (b >= B) return;
    float x[D];
    bool inside = true;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        x[d] = widen(inputs[(size_t)b * D + d]);
        if (x[d] < 0 || x[d] > 1) inside = false;
    }
    h16* out = outputs + ((size_t)level * B + b) * C;
    h16* g_out = dy_dx ? dy_dx + ((size_t)b * L + level) * (D * C) : nullptr;
    if (!inside) {
#pragma unroll
        for (int c = 0; c < C; ++c) out[c] = H(0.0f);
        if (g_out)
#pragma unroll
            for (int i = 0; i < D * C; ++i) g_out[i] = H(0.0f);
        return;
    }
    const uint32_t row0 = (uint32_t)offsets[level];
    const uint32_t size = (uint32_t)offsets[level + 1] - row0;
    const uint32_t res = ls.resolution[level];
    const LevelGeom<D> g = SMOOTH ? make_level_geom<D>(size, res, true) : make_level_geom<D>(size, align_corners ? res : res + 1, gridtype == 0);
    const float scale = ls.scale[level];
    const float offset = SMOOTH ? 0.0f : (align_corners ? 0.0f : 0.5f);
    const h16* table = embeddings + (size_t)row0 * C;

    float pos[D], dpos[D];
    uint32_t cell[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float p = x[d] * scale + offset;
        cell[d] = (uint32_t)floorf(p);
        p -= (float)cell[d];
        if constexpr (SMOOTH) { dpos[d] = 6 * p * (1.0f - p); pos[d] = p * p * (3.0f - 2.0f * p); }
        else { dpos[d] = 1.0f; pos[d] = p; }
    }
    h16 result[C];
#pragma unroll
    for (int c = 0; c < C; ++c) result[c] = H(0.0f);
#pragma unroll
    for (int i = 0; i < (1 << D); ++i) {
        float w = 1;
        uint32_t q[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (((i >> d) & 1) == 0) { w *= 1 - pos[d]; q[d] = cell[d]; }
            else { w *= pos[d]; q[d] = cell[d] + 1; }
        }
        const uint32_t row = cell_row<D>(g, q);
#pragma unroll
        for (int c = 0; c < C; ++c) result[c] = add_f(result[c], w * F(table[(size_t)row * C + c]));
    }
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = result[c];
    if (!g_out) return;
#pragma unroll
    for (int gd = 0; gd < D; ++gd) {
        h16 acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = H(0.0f);
#pragma unroll
        for (int j = 0; j < (1 << (D - 1)); ++j) {
            float w = scale;
            uint32_t q[D];
#pragma unroll
            for (int nd = 0; nd < D - 1; ++nd) {
                const int d = nd >= gd ? nd + 1 : nd;
                if (((j >> nd) & 1) == 0) { w *= 1 - pos[d]; q[d] = cell[d]; }
                else { w *= pos[d]; q[d] = cell[d] + 1; }
            }
            q[gd] = cell[gd];
            const uint32_t left = cell_row<D>(g, q);
            q[gd] = cell[gd] + 1;
            const uint32_t right = cell_row<D>(g, q);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const h16 diff = H(F(table[(size_t)right * C + c]) - F(table[(size_t)left * C + c]));
                const float t = SMOOTH ? w * F(diff) * dpos[gd] : w * F(diff);
                acc[c] = add_f(acc[c], t);
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) g_out[gd * C + c] = acc[c];
    }
}

// fp16 atomic add of one value: compare-and-swap on the 32-bit word that holds it (C == 1 tables only; the reference's own
// helper for this case is "very slow ... never used", hashencoder.cu:24-28)
__device__ __forceinline__ void atomic_add_h(h16* addr, h16 v) {
    uint32_t* word = reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(addr) & ~(uintptr_t)3);
    const bool upper = (reinterpret_cast<uintptr_t>(addr) & 2) != 0;
    uint32_t old = *word, assumed;
    do {
        assumed = old;
        const uint16_t cur = upper ? (uint16_t)(assumed >> 16) : (uint16_t)(assumed & 0xffffu);
        const h16 sum = H(F(__builtin_bit_cast(h16, cur)) + F(v));
        const uint32_t bits = (uint32_t)__builtin_bit_cast(uint16_t, sum);
        const uint32_t next = upper ? ((assumed & 0x0000ffffu) | (bits << 16)) : ((assumed & 0xffff0000u) | bits);
        old = atomicCAS(word, assumed, next);
    } while (old != assumed);
}
// two adjacent values (an even channel pair, 4-byte aligned): one packed fp16 atomic, the reference's __half2 path
__device__ __forceinline__ void atomic_add_h2(h16* addr, h16 a, h16 b) {
    typedef __attribute__((address_space(1))) h16x2 global_h2;
    const h16x2 v = {a, b};
    (void)__builtin_amdgcn_global_atomic_fadd_v2f16((global_h2*)addr, v);
}

// kernel_grid_backward<at::Half, D, C, N_C> (hashencoder.cu:257-343, gridencoder.cu:226-305)
template <int D, int C, bool SMOOTH, typename IN>
__global__ void __launch_bounds__(kBlock) k_grid_backward_table_h(const h16* __restrict__ grad, const IN* __restrict__ inputs,
                                                                  const int32_t* __restrict__ offsets, h16* __restrict__ grad_table,
                                                                  uint32_t B, uint32_t L, LevelScale ls, uint32_t gridtype, bool align_corners) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t level = blockIdx.y;
    if (b >= B) return;
    float x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        x[d] = widen(inputs[(size_t)b * D + d]);
        if (x[d] < 0 || x[d] > 1) return;
    }
    const uint32_t row0 = (uint32_t)offsets[level];
    const uint32_t size = (uint32_t)offsets[level + 1] - row0;
    const uint32_t res = ls.resolution[level];
    const LevelGeom<D> g = SMOOTH ? make_level_geom<D>(size, res, true) : make_level_geom<D>(size, align_corners ? res : res + 1, gridtype == 0);
    const float scale = ls.scale[level];
    const float offset = SMOOTH ? 0.0f : (align_corners ? 0.0f : 0.5f);
    float pos[D];
    uint32_t cell[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float p = x[d] * scale + offset;
        cell[d] = (uint32_t)floorf(p);
        p -= (float)cell[d];
        pos[d] = SMOOTH ? p * p * (3.0f - 2.0f * p) : p;
    }
    h16 gcur[C];
#pragma unroll
    for (int c = 0; c < C; ++c) gcur[c] = grad[((size_t)level * B + b) * C + c];
    h16* t = grad_table + (size_t)row0 * C;
#pragma unroll
    for (int i = 0; i < (1 << D); ++i) {
        float w = 1;
        uint32_t q[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (((i >> d) & 1) == 0) { w *= 1 - pos[d]; q[d] = cell[d]; }
            else { w *= pos[d]; q[d] = cell[d] + 1; }
        }
        const uint32_t row = cell_row<D>(g, q);
        if constexpr (C % 2 == 0) {
#pragma unroll
            for (int c = 0; c < C; c += 2) atomic_add_h2(t + (size_t)row * C + c, H(w * F(gcur[c])), H(w * F(gcur[c + 1])));
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) atomic_add_h(t + (size_t)row * C + c, H(w * F(gcur[c])));
        }
    }
}

// kernel_input_backward<at::Half, D, C> (hashencoder.cu:346-372, gridencoder.cu:308-335)
template <int D, int C>
__global__ void __launch_bounds__(kBlock) k_grid_input_backward_h(const h16* __restrict__ grad, const h16* __restrict__ dy_dx,
                                                                  h16* __restrict__ grad_inputs, uint32_t B, uint32_t L) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const h16* dy = dy_dx + (size_t)b * L * D * C;
    h16 acc = H(0.0f);
    for (uint32_t l = 0; l < L; ++l)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const h16 prod = H(F(grad[((size_t)l * B + b) * C + c]) * F(dy[(size_t)l * D * C + d * C + c]));
            acc = H(F(acc) + F(prod));
        }
    grad_inputs[t] = acc;
}

// kernel_grid_second_backward_grad<at::Half, D, C, N_C> (hashencoder.cu:375-428): a Half product and a Half sum per term
template <int D, int C>
__global__ void __launch_bounds__(kBlock) k_second_backward_grad_h(const h16* __restrict__ ggx, const h16* __restrict__ dy_dx,
                                                                   h16* __restrict__ grad_grad, uint32_t B, uint32_t L) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t level = blockIdx.y;
    if (b >= B) return;
    const h16* j = dy_dx + ((size_t)b * L + level) * (D * C);
    h16 r[C];
#pragma unroll
    for (int c = 0; c < C; ++c) r[c] = H(0.0f);
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int c = 0; c < C; ++c) r[c] = H(F(r[c]) + F(H(F(ggx[(size_t)b * D + d]) * F(j[d * C + c]))));
#pragma unroll
    for (int c = 0; c < C; ++c) grad_grad[((size_t)level * B + b) * C + c] = r[c];
}

// kernel_grid_second_backward_embedding<at::Half, D, C, N_C> (hashencoder.cu:431-595): the corner cache is Half -- every
// `cache +-= w * grad * ggx[gd] * smoothstep'` narrows the float product, then adds / subtracts in Half -- and is scattered
// with packed fp16 atomics
template <int D, int C>
__global__ void __launch_bounds__(kBlock) k_second_backward_table_h(const h16* __restrict__ grad, const h16* __restrict__ inputs,
                                                                    const int32_t* __restrict__ offsets, const h16* __restrict__ ggx,
                                                                    h16* __restrict__ grad2_table, uint32_t B, uint32_t L, LevelScale ls) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t level = blockIdx.y;
    if (b >= B) return;
    float x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        x[d] = F(inputs[(size_t)b * D + d]);
        if (x[d] < 0 || x[d] > 1) return;
    }
    const uint32_t row0 = (uint32_t)offsets[level];
    const uint32_t size = (uint32_t)offsets[level + 1] - row0;
    const LevelGeom<D> g = make_level_geom<D>(size, ls.resolution[level], true);
    const float scale = ls.scale[level];
    float w1[D], dw[D];
    uint32_t cell[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float p = x[d] * scale;
        cell[d] = (uint32_t)floorf(p);
        p -= (float)cell[d];
        dw[d] = 6 * p * (1.0f - p);
        w1[d] = p * p * (3.0f - 2.0f * p);
    }
    h16 gcur[C], gg[D];
#pragma unroll
    for (int c = 0; c < C; ++c) gcur[c] = grad[((size_t)level * B + b) * C + c];
#pragma unroll
    for (int d = 0; d < D; ++d) gg[d] = ggx[(size_t)b * D + d];
    h16 corner[1 << D][C];
#pragma unroll
    for (int i = 0; i < (1 << D); ++i)
#pragma unroll
        for (int c = 0; c < C; ++c) corner[i][c] = H(0.0f);
#pragma unroll
    for (int gd = 0; gd < D; ++gd) {
#pragma unroll
        for (int j = 0; j < (1 << (D - 1)); ++j) {
            float w = scale;
            int lo = 0;
#pragma unroll
            for (int nd = 0; nd < D - 1; ++nd) {
                const int d = nd >= gd ? nd + 1 : nd;
                const int bit = (j >> nd) & 1;
                w *= bit ? w1[d] : 1 - w1[d];
                lo |= bit << d;
            }
            const int hi = lo | (1 << gd);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const h16 v = H(w * F(gcur[c]) * F(gg[gd]) * dw[gd]);
                corner[hi][c] = H(F(corner[hi][c]) + F(v));
                corner[lo][c] = H(F(corner[lo][c]) - F(v));
            }
        }
    }
    h16* t = grad2_table + (size_t)row0 * C;
#pragma unroll
    for (int i = 0; i < (1 << D); ++i) {
        uint32_t q[D];
#pragma unroll
        for (int d = 0; d < D; ++d) q[d] = cell[d] + ((i >> d) & 1);
        const uint32_t row = cell_row<D>(g, q);
#pragma unroll
        for (int c = 0; c < C; c += 2) atomic_add_h2(t + (size_t)row * C + c, corner[i][c], corner[i][c + 1]);
    }
}

// (the dimension range is a COMPILE-time bound: with a run-time one every caller instantiated its kernels for D = 1 .. 5, 42 of them
//  unreachable through the C ABI -- the hash encoder's entry points take D = 2, 3 only)
template <int D_LO, int D_HI, typename Fn>
int dispatch_dc_h(uint32_t D, uint32_t C, const char* who, Fn&& fn) {
#define ENVIDR_CASE(DD, CC) if (D == DD && C == CC) return fn(std::integral_constant<int, DD>{}, std::integral_constant<int, CC>{});
#define ENVIDR_ROW(DD) if constexpr (DD >= D_LO && DD <= D_HI) { ENVIDR_CASE(DD, 1) ENVIDR_CASE(DD, 2) ENVIDR_CASE(DD, 4) ENVIDR_CASE(DD, 8) }
    ENVIDR_ROW(1) ENVIDR_ROW(2) ENVIDR_ROW(3) ENVIDR_ROW(4) ENVIDR_ROW(5)
#undef ENVIDR_ROW
#undef ENVIDR_CASE
    set_error("%s: unsupported (D=%u, C=%u); D must be %d..%d and C one of 1, 2, 4, 8", who, D, C, D_LO, D_HI);
    return ENVIDR_EINVAL;
}

template <bool SMOOTH, typename IN>
int forward_h(const IN* inputs, const uint16_t* embeddings, const int32_t* offsets, uint16_t* outputs, uint32_t B, uint32_t D, uint32_t C,
              uint32_t L, float S, uint32_t H_, uint16_t* dy_dx, uint32_t gridtype, int align_corners, envidr_stream_t stream, const char* who) {
    ENVIDR_REQUIRE(L >= 1 && L <= (uint32_t)kMaxLevels, "%s: L=%u out of range [1,%d]", who, L, kMaxLevels);
    ENVIDR_REQUIRE(gridtype <= 1, "%s: gridtype must be 0 (hash) or 1 (tiled)", who);
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(inputs && embeddings && offsets && outputs, "%s: null pointer", who);
    const LevelScale ls = make_level_scale(L, S, H_);
    const dim3 grid(ceil_div(B, kBlock), L);
    return dispatch_dc_h<(SMOOTH ? 2 : 1), (SMOOTH ? 3 : 5)>(D, C, who, [&](auto d, auto c) {
        constexpr int DD = decltype(d)::value, CC = decltype(c)::value;
        hipLaunchKernelGGL((k_grid_forward_h<DD, CC, SMOOTH, IN>), grid, dim3(kBlock), 0, as_stream(stream), inputs,
                           reinterpret_cast<const h16*>(embeddings), offsets, reinterpret_cast<h16*>(outputs), B, L, ls,
                           reinterpret_cast<h16*>(dy_dx), gridtype, align_corners != 0);
        return check_launch("k_grid_forward_h");
    });
}

template <bool SMOOTH, typename IN>
int backward_h(const uint16_t* grad, const IN* inputs, const int32_t* offsets, uint16_t* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
               uint32_t L, float S, uint32_t H_, const uint16_t* dy_dx, uint16_t* grad_inputs, uint32_t gridtype, int align_corners,
               envidr_stream_t stream, const char* who) {
    ENVIDR_REQUIRE(L >= 1 && L <= (uint32_t)kMaxLevels, "%s: L=%u out of range", who, L);
    ENVIDR_REQUIRE(gridtype <= 1, "%s: gridtype must be 0 (hash) or 1 (tiled)", who);
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(grad && inputs && offsets, "%s: null pointer", who);
    ENVIDR_REQUIRE(!dy_dx || grad_inputs, "%s: dy_dx given but grad_inputs is null", who);
    const LevelScale ls = make_level_scale(L, S, H_);
    return dispatch_dc_h<(SMOOTH ? 2 : 1), (SMOOTH ? 3 : 5)>(D, C, who, [&](auto d, auto c) {
        constexpr int DD = decltype(d)::value, CC = decltype(c)::value;
        int rc = ENVIDR_OK;
        if (grad_embeddings) {
            hipLaunchKernelGGL((k_grid_backward_table_h<DD, CC, SMOOTH, IN>), dim3(ceil_div(B, kBlock), L), dim3(kBlock), 0, as_stream(stream),
                               reinterpret_cast<const h16*>(grad), inputs, offsets, reinterpret_cast<h16*>(grad_embeddings), B, L, ls, gridtype,
                               align_corners != 0);
            rc = check_launch("k_grid_backward_table_h");
            if (rc) return rc;
        }
        if (dy_dx) {
            hipLaunchKernelGGL((k_grid_input_backward_h<DD, CC>), dim3(ceil_div(B * DD, kBlock)), dim3(kBlock), 0, as_stream(stream),
                               reinterpret_cast<const h16*>(grad), reinterpret_cast<const h16*>(dy_dx), reinterpret_cast<h16*>(grad_inputs), B, L);
            rc = check_launch("k_grid_input_backward_h");
        }
        return rc;
    });
}

}  // namespace

extern "C" {

int envidr_hash_encode_forward_f16(const uint16_t* inputs, const uint16_t* embeddings, const int32_t* offsets, uint16_t* outputs, uint32_t B,
                                   uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, uint16_t* dy_dx,
                                   envidr_stream_t stream) {
    ENVIDR_REQUIRE(!calc_grad_inputs || dy_dx || B == 0, "hash_encode_forward_f16: dy_dx is null but calc_grad_inputs is set");
    return forward_h<true>(reinterpret_cast<const h16*>(inputs), embeddings, offsets, outputs, B, D, C, L, S, H, calc_grad_inputs ? dy_dx : nullptr, 0, 0,
                           stream, "hash_encode_forward_f16");
}

int envidr_hash_encode_backward_f16(const uint16_t* grad, const uint16_t* inputs, const uint16_t* embeddings, const int32_t* offsets,
                                    uint16_t* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                    int calc_grad_inputs, const uint16_t* dy_dx, uint16_t* grad_inputs, envidr_stream_t stream) {
    (void)embeddings;
    ENVIDR_REQUIRE(!calc_grad_inputs || (dy_dx && grad_inputs) || B == 0, "hash_encode_backward_f16: dy_dx/grad_inputs null");
    return backward_h<true>(grad, reinterpret_cast<const h16*>(inputs), offsets, grad_embeddings, B, D, C, L, S, H, calc_grad_inputs ? dy_dx : nullptr,
                            grad_inputs, 0, 0, stream, "hash_encode_backward_f16");
}

int envidr_hash_encode_second_backward_f16(const uint16_t* grad, const uint16_t* inputs, const uint16_t* embeddings, const int32_t* offsets,
                                           uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H_, int calc_grad_inputs,
                                           const uint16_t* dy_dx, const uint16_t* grad_grad_inputs, uint16_t* grad_grad, uint16_t* grad2_embeddings,
                                           envidr_stream_t stream) {
    (void)embeddings; (void)calc_grad_inputs;
    const char* who = "hash_encode_second_backward_f16";
    ENVIDR_REQUIRE(L >= 1 && L <= (uint32_t)kMaxLevels, "%s: L=%u out of range", who, L);
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(grad && inputs && offsets && dy_dx && grad_grad_inputs && grad_grad && grad2_embeddings, "%s: null pointer", who);
    ENVIDR_REQUIRE(C != 1, "%s: C=1 is not supported (reference: hashencoder.cu:673-679)", who);
    const LevelScale ls = make_level_scale(L, S, H_);
    return dispatch_dc_h<2, 3>(D, C, who, [&](auto d, auto c) {
        constexpr int DD = decltype(d)::value, CC = decltype(c)::value;
        if constexpr (CC == 1) return (int)ENVIDR_EINVAL;
        else {
            const dim3 grid(ceil_div(B, kBlock), L);
            hipLaunchKernelGGL((k_second_backward_grad_h<DD, CC>), grid, dim3(kBlock), 0, as_stream(stream), reinterpret_cast<const h16*>(grad_grad_inputs),
                               reinterpret_cast<const h16*>(dy_dx), reinterpret_cast<h16*>(grad_grad), B, L);
            int rc = check_launch("k_second_backward_grad_h");
            if (rc) return rc;
            hipLaunchKernelGGL((k_second_backward_table_h<DD, CC>), grid, dim3(kBlock), 0, as_stream(stream), reinterpret_cast<const h16*>(grad),
                               reinterpret_cast<const h16*>(inputs), offsets, reinterpret_cast<const h16*>(grad_grad_inputs),
                               reinterpret_cast<h16*>(grad2_embeddings), B, L, ls);
            return check_launch("k_second_backward_table_h");
        }
    });
}

int envidr_grid_encode_forward_f16(const float* inputs, const uint16_t* embeddings, const int32_t* offsets, uint16_t* outputs, uint32_t B,
                                   uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint16_t* dy_dx, uint32_t gridtype, int align_corners,
                                   envidr_stream_t stream) {
    return forward_h<false>(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, stream, "grid_encode_forward_f16");
}

int envidr_grid_encode_backward_f16(const uint16_t* grad, const float* inputs, const uint16_t* embeddings, const int32_t* offsets,
                                    uint16_t* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                    const uint16_t* dy_dx, uint16_t* grad_inputs, uint32_t gridtype, int align_corners, envidr_stream_t stream) {
    (void)embeddings;
    ENVIDR_REQUIRE(grad_embeddings || B == 0, "grid_encode_backward_f16: null pointer");
    return backward_h<false>(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners, stream,
                             "grid_encode_backward_f16");
}

}  // extern "C"
