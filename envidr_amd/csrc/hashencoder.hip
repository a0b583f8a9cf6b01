// hashencoder operators for gfx950 (smoothstep multiresolution hash grid with analytic input
// derivative) -- replaces hashencoder/src/hashencoder.cu (hash_encode_forward :725,
// hash_encode_backward :762, hash_encode_second_backward :795; declarations hashencoder.h:13-15).
//
// Work decomposition: one lane per (point, level).  Workgroups are mapped so that each XCD works
// on one level at a time (level = f(block % 8)): a hashed level's table is 4 MiB, exactly one
// XCD's L2, so the random 8-byte gathers of that level stay L2-resident on the XCD that owns it
// instead of every XCD thrashing all 16 tables.
#include "grid_core.hip.h"

using namespace envidr;

namespace {

constexpr uint32_t kXcds = 8;

// block -> (level, chunk) such that blocks resident on one XCD (block % 8) sweep one level after
// another.  Returns false for padding blocks.
__device__ __forceinline__ bool xcd_level_chunk(uint32_t L, uint32_t chunks, uint32_t& level, uint32_t& chunk) {
    const uint32_t xcd = blockIdx.x % kXcds;
    const uint32_t j = blockIdx.x / kXcds;          // sequence number inside this XCD's queue
    const uint32_t k = j / chunks;                  // which of this XCD's levels
    level = xcd + kXcds * k;
    chunk = j - k * chunks;
    return level < L;
}
inline uint32_t xcd_grid_blocks(uint32_t L, uint32_t chunks) { return kXcds * ceil_div(L, kXcds) * chunks; }

template <int D>
__device__ __forceinline__ bool load_point(const float* __restrict__ inputs, uint32_t b, float (&x)[D]) {
    bool inside = true;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        x[d] = inputs[(size_t)b * D + d];
        if (x[d] < 0 || x[d] > 1) inside = false;
    }
    return inside;
}

// ------------------------------------------------------------------------------------------
// forward (+ optional dy_dx)
// ------------------------------------------------------------------------------------------
template <int D, int C, bool GRAD>
__global__ void __launch_bounds__(kBlock) k_hash_forward(const float* __restrict__ inputs,
                                                         const float* __restrict__ embeddings,
                                                         const int32_t* __restrict__ offsets,
                                                         float* __restrict__ outputs, uint32_t B, uint32_t L,
                                                         LevelScale ls, uint32_t chunks, float* __restrict__ dy_dx) {
    uint32_t level, chunk;
    if (!xcd_level_chunk(L, chunks, level, chunk)) return;
    const uint32_t b = chunk * blockDim.x + threadIdx.x;
    if (b >= B) return;

    const uint32_t row0 = (uint32_t)offsets[level];
    const uint32_t size = (uint32_t)offsets[level + 1] - row0;
    const LevelGeom<D> g = make_level_geom<D>(size, ls.resolution[level], /*allow_hash=*/true);

    float x[D], out[C], grad[D][C];
    const bool inside = load_point<D>(inputs, b, x);
    if (inside) {
        eval_level<D, C, /*SMOOTH=*/true, GRAD>(x, embeddings + (size_t)row0 * C, g, ls.scale[level], 0.0f, out, grad);
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) out[c] = 0;
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int c = 0; c < C; ++c) grad[d][c] = 0;
    }

    float* o = outputs + ((size_t)level * B + b) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = out[c];
    if constexpr (GRAD) {
        float* g_out = dy_dx + ((size_t)b * L + level) * (D * C);
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int c = 0; c < C; ++c) g_out[d * C + c] = grad[d][c];
    }
}

// ------------------------------------------------------------------------------------------
// backward w.r.t. the table: scatter w * grad into the 2^D corner rows (hardware fp32 atomics)
// ------------------------------------------------------------------------------------------
template <int D, int C>
__global__ void __launch_bounds__(kBlock) k_hash_backward_table(const float* __restrict__ grad,
                                                                const float* __restrict__ inputs,
                                                                const int32_t* __restrict__ offsets,
                                                                float* __restrict__ grad_table, uint32_t B, uint32_t L,
                                                                LevelScale ls, uint32_t chunks) {
    uint32_t level, chunk;
    if (!xcd_level_chunk(L, chunks, level, chunk)) return;
    const uint32_t b = chunk * blockDim.x + threadIdx.x;
    // (no early exit per lane: the lanes of a wave combine their adds, RunScatter)
    float x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = 0.5f;
    const bool on = b < B && load_point<D>(inputs, b, x);   // outside the cube: nothing to add (the table gradient starts at zero)
    if (!on) {
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = 0.5f;
    }

    const uint32_t row0 = (uint32_t)offsets[level];
    const uint32_t size = (uint32_t)offsets[level + 1] - row0;
    const LevelGeom<D> g = make_level_geom<D>(size, ls.resolution[level], true);
    const float scale = ls.scale[level];

    float w1[D];
    uint32_t cell[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float p = x[d] * scale;
        cell[d] = (uint32_t)floorf(p);
        p -= (float)cell[d];
        w1[d] = p * p * (3.0f - 2.0f * p);
    }
    float gcur[C];
#pragma unroll
    for (int c = 0; c < C; ++c) gcur[c] = on ? grad[((size_t)level * B + b) * C + c] : 0.0f;

    float* t = grad_table + (size_t)row0 * C;
    const RunScatter rs(on);
    bool combine = false;
#pragma unroll
    for (int i = 0; i < (1 << D); ++i) {
        float w = 1;
        uint32_t q[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int bit = (i >> d) & 1;
            w *= bit ? w1[d] : 1 - w1[d];
            q[d] = cell[d] + bit;
        }
        const uint32_t row = cell_row<D>(g, q);
        if (i == 0) combine = rs.worth(row, on);
        float v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = w * gcur[c];
        rs.template add<C>(t, row, on, v, combine);
    }
}

// grad_inputs[b,d] = sum_{l,c} grad[l,b,c] * dy_dx[b,l,d,c]      (hashencoder.cu:346-372)
template <int D, int C>
__global__ void __launch_bounds__(kBlock) k_input_backward(const float* __restrict__ grad,
                                                           const float* __restrict__ dy_dx,
                                                           float* __restrict__ grad_inputs, uint32_t B, uint32_t L) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float* j = dy_dx + (size_t)b * L * D * C + d * C;
    float acc = 0;
    for (uint32_t l = 0; l < L; ++l) {
#pragma unroll
        for (int c = 0; c < C; ++c) acc += grad[((size_t)l * B + b) * C + c] * j[(size_t)l * D * C + c];
    }
    grad_inputs[t] = acc;
}

// grad_grad[l,b,c] = sum_d ggx[b,d] * dy_dx[b,l,d,c]              (hashencoder.cu:375-428)
template <int D, int C>
__global__ void __launch_bounds__(kBlock) k_second_backward_grad(const float* __restrict__ ggx,
                                                                 const float* __restrict__ dy_dx,
                                                                 float* __restrict__ grad_grad, uint32_t B,
                                                                 uint32_t L) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t level = blockIdx.y;
    if (b >= B) return;
    const float* j = dy_dx + ((size_t)b * L + level) * (D * C);
    float r[C];
#pragma unroll
    for (int c = 0; c < C; ++c) r[c] = 0;
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int c = 0; c < C; ++c) r[c] += ggx[(size_t)b * D + d] * j[d * C + c];
#pragma unroll
    for (int c = 0; c < C; ++c) grad_grad[((size_t)level * B + b) * C + c] = r[c];
}

// table second gradient: +/- w * grad * ggx[gd] * smoothstep'(frac_gd) on the corner pairs
// along gd, accumulated per corner in registers then scattered      (hashencoder.cu:431-595)
template <int D, int C>
__global__ void __launch_bounds__(kBlock) k_second_backward_table(const float* __restrict__ grad,
                                                                  const float* __restrict__ inputs,
                                                                  const int32_t* __restrict__ offsets,
                                                                  const float* __restrict__ ggx,
                                                                  float* __restrict__ grad2_table, uint32_t B,
                                                                  uint32_t L, LevelScale ls, uint32_t chunks) {
    uint32_t level, chunk;
    if (!xcd_level_chunk(L, chunks, level, chunk)) return;
    const uint32_t b = chunk * blockDim.x + threadIdx.x;
    float x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = 0.5f;
    const bool on = b < B && load_point<D>(inputs, b, x);          // (no early exit per lane: RunScatter combines across the wave)
    if (!on) {
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = 0.5f;
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
    float gcur[C], gg[D];
#pragma unroll
    for (int c = 0; c < C; ++c) gcur[c] = on ? grad[((size_t)level * B + b) * C + c] : 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) gg[d] = on ? ggx[(size_t)b * D + d] : 0.0f;

    float corner[1 << D][C];
#pragma unroll
    for (int i = 0; i < (1 << D); ++i)
#pragma unroll
        for (int c = 0; c < C; ++c) corner[i][c] = 0;

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
                const float v = w * gcur[c] * gg[gd] * dw[gd];
                corner[hi][c] += v;
                corner[lo][c] -= v;
            }
        }
    }

    float* t = grad2_table + (size_t)row0 * C;
    const RunScatter rs(on);
    bool combine = false;
#pragma unroll
    for (int i = 0; i < (1 << D); ++i) {
        uint32_t q[D];
#pragma unroll
        for (int d = 0; d < D; ++d) q[d] = cell[d] + ((i >> d) & 1);
        const uint32_t row = cell_row<D>(g, q);
        if (i == 0) combine = rs.worth(row, on);
        rs.template add<C>(t, row, on, corner[i], combine);
    }
}

// ------------------------------------------------------------------------------------------
// dispatch helpers
// ------------------------------------------------------------------------------------------
template <typename F>
int dispatch_dc(uint32_t D, uint32_t C, const char* who, F&& f) {
#define ENVIDR_CASE(DD, CC) \
    if (D == DD && C == CC) return f(std::integral_constant<int, DD>{}, std::integral_constant<int, CC>{});
    ENVIDR_CASE(2, 1) ENVIDR_CASE(2, 2) ENVIDR_CASE(2, 4) ENVIDR_CASE(2, 8)
    ENVIDR_CASE(3, 1) ENVIDR_CASE(3, 2) ENVIDR_CASE(3, 4) ENVIDR_CASE(3, 8)
#undef ENVIDR_CASE
    set_error("%s: unsupported (D=%u, C=%u); D must be 2 or 3 and C one of 1, 2, 4, 8", who, D, C);
    return ENVIDR_EINVAL;
}

}  // namespace

extern "C" {

int envidr_hash_encode_forward(const float* inputs, const float* embeddings, const int32_t* offsets, float* outputs,
                               uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                               int calc_grad_inputs, float* dy_dx, envidr_stream_t stream) {
    ENVIDR_REQUIRE(L >= 1 && L <= (uint32_t)kMaxLevels, "hash_encode_forward: L=%u out of range [1,%d]", L, kMaxLevels);
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(inputs && embeddings && offsets && outputs, "hash_encode_forward: null pointer");
    ENVIDR_REQUIRE(!calc_grad_inputs || dy_dx, "hash_encode_forward: dy_dx is null but calc_grad_inputs is set");
    const LevelScale ls = make_level_scale(L, S, H);
    const uint32_t chunks = ceil_div(B, kBlock);
    const dim3 grid(xcd_grid_blocks(L, chunks));
    return dispatch_dc(D, C, "hash_encode_forward", [&](auto d, auto c) {
        constexpr int DD = decltype(d)::value, CC = decltype(c)::value;
        if (calc_grad_inputs)
            hipLaunchKernelGGL((k_hash_forward<DD, CC, true>), grid, dim3(kBlock), 0, as_stream(stream), inputs,
                               embeddings, offsets, outputs, B, L, ls, chunks, dy_dx);
        else
            hipLaunchKernelGGL((k_hash_forward<DD, CC, false>), grid, dim3(kBlock), 0, as_stream(stream), inputs,
                               embeddings, offsets, outputs, B, L, ls, chunks, dy_dx);
        return check_launch("k_hash_forward");
    });
}

int envidr_hash_encode_backward(const float* grad, const float* inputs, const float* embeddings,
                                const int32_t* offsets, float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                uint32_t L, float S, uint32_t H, int calc_grad_inputs, const float* dy_dx,
                                float* grad_inputs, envidr_stream_t stream) {
    (void)embeddings;
    ENVIDR_REQUIRE(L >= 1 && L <= (uint32_t)kMaxLevels, "hash_encode_backward: L=%u out of range", L);
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(grad && inputs && offsets, "hash_encode_backward: null pointer");
    ENVIDR_REQUIRE(!calc_grad_inputs || (dy_dx && grad_inputs), "hash_encode_backward: dy_dx/grad_inputs null");
    const LevelScale ls = make_level_scale(L, S, H);
    const uint32_t chunks = ceil_div(B, kBlock);
    return dispatch_dc(D, C, "hash_encode_backward", [&](auto d, auto c) {
        constexpr int DD = decltype(d)::value, CC = decltype(c)::value;
        if (grad_embeddings) {
            hipLaunchKernelGGL((k_hash_backward_table<DD, CC>), dim3(xcd_grid_blocks(L, chunks)), dim3(kBlock), 0,
                               as_stream(stream), grad, inputs, offsets, grad_embeddings, B, L, ls, chunks);
            const int rc = check_launch("k_hash_backward_table");
            if (rc) return rc;
        }
        if (calc_grad_inputs) {
            hipLaunchKernelGGL((k_input_backward<DD, CC>), dim3(ceil_div(B * DD, kBlock)), dim3(kBlock), 0,
                               as_stream(stream), grad, dy_dx, grad_inputs, B, L);
            return check_launch("k_input_backward");
        }
        return ENVIDR_OK;
    });
}

int envidr_hash_encode_second_backward(const float* grad, const float* inputs, const float* embeddings,
                                       const int32_t* offsets, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                       float S, uint32_t H, int calc_grad_inputs, const float* dy_dx,
                                       const float* grad_grad_inputs, float* grad_grad, float* grad2_embeddings,
                                       envidr_stream_t stream) {
    (void)embeddings; (void)calc_grad_inputs;
    ENVIDR_REQUIRE(L >= 1 && L <= (uint32_t)kMaxLevels, "hash_encode_second_backward: L=%u out of range", L);
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(grad && inputs && offsets && dy_dx && grad_grad_inputs && grad_grad && grad2_embeddings,
                   "hash_encode_second_backward: null pointer");
    ENVIDR_REQUIRE(C != 1, "hash_encode_second_backward: C=1 is not supported (reference: hashencoder.cu:673-679)");
    const LevelScale ls = make_level_scale(L, S, H);
    const uint32_t chunks = ceil_div(B, kBlock);
    return dispatch_dc(D, C, "hash_encode_second_backward", [&](auto d, auto c) {
        constexpr int DD = decltype(d)::value, CC = decltype(c)::value;
        hipLaunchKernelGGL((k_second_backward_grad<DD, CC>), dim3(chunks, L), dim3(kBlock), 0, as_stream(stream),
                           grad_grad_inputs, dy_dx, grad_grad, B, L);
        int rc = check_launch("k_second_backward_grad");
        if (rc) return rc;
        hipLaunchKernelGGL((k_second_backward_table<DD, CC>), dim3(xcd_grid_blocks(L, chunks)), dim3(kBlock), 0,
                           as_stream(stream), grad, inputs, offsets, grad_grad_inputs, grad2_embeddings, B, L, ls,
                           chunks);
        return check_launch("k_second_backward_table");
    });
}

}  // extern "C"
