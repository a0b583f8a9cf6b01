// gridencoder operators for gfx950 (torch-ngp's stock hash / tiled grid: linear interpolation,
// +0.5 cell offset unless align_corners, dense stride res+1) -- replaces
// gridencoder/src/gridencoder.cu (grid_encode_forward :423, grid_encode_backward :452;
// declarations gridencoder.h:12-13).  fp32 tables only: the reference switches to fp16 tables only
// under autocast, which none of its shipped configs enables.
//
// Same XCD-aware level scheduling as hashencoder.hip; the per-point math is grid_core.hip.h's
// eval_level<SMOOTH=false>.
#include "grid_core.hip.h"
#include "input_rows.hip.h"

using namespace envidr;

namespace {

constexpr uint32_t kXcds = 8;

__device__ __forceinline__ bool xcd_level_chunk(uint32_t L, uint32_t chunks, uint32_t& level, uint32_t& chunk) {
    const uint32_t xcd = blockIdx.x % kXcds;
    const uint32_t j = blockIdx.x / kXcds;
    const uint32_t k = j / chunks;
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

template <int D, int C, bool GRAD>
__global__ void __launch_bounds__(kBlock) k_grid_forward(const float* __restrict__ inputs,
                                                         const float* __restrict__ embeddings,
                                                         const int32_t* __restrict__ offsets,
                                                         float* __restrict__ outputs, uint32_t B, uint32_t L,
                                                         LevelScale ls, uint32_t chunks, float* __restrict__ dy_dx,
                                                         uint32_t gridtype, bool align_corners) {
    // level-major block order, like hashencoder.hip's k_hash_forward (the level-per-XCD pinning of rounds 1-4 lost to it on every input
    // order when timed against the reference's kernel compiled for this GPU: 3.84 -> 3.4 ms on random points, 2.47 -> 1.7 ms ray-major)
    const uint32_t level = blockIdx.x / chunks, chunk = blockIdx.x - level * chunks;
    const uint32_t b = chunk * blockDim.x + threadIdx.x;
    if (b >= B) return;

    const uint32_t row0 = (uint32_t)offsets[level];
    const uint32_t size = (uint32_t)offsets[level + 1] - row0;
    const uint32_t res = ls.resolution[level];
    const LevelGeom<D> g = make_level_geom<D>(size, align_corners ? res : res + 1, gridtype == 0);

    float x[D], out[C], grad[D][C];
    if (load_point<D>(inputs, b, x)) {
        eval_level<D, C, /*SMOOTH=*/false, GRAD>(x, embeddings + (size_t)row0 * C, g, ls.scale[level],
                                                 align_corners ? 0.0f : 0.5f, out, grad);
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

template <int D, int C>
__global__ void __launch_bounds__(kBlock) k_grid_backward_table(const float* __restrict__ grad,
                                                                const float* __restrict__ inputs,
                                                                const int32_t* __restrict__ offsets,
                                                                float* __restrict__ grad_table, uint32_t B, uint32_t L,
                                                                LevelScale ls, uint32_t chunks, uint32_t gridtype,
                                                                bool align_corners) {
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
    const uint32_t res = ls.resolution[level];
    const LevelGeom<D> g = make_level_geom<D>(size, align_corners ? res : res + 1, gridtype == 0);
    const float scale = ls.scale[level], off = align_corners ? 0.0f : 0.5f;

    float w1[D];
    uint32_t cell[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float p = x[d] * scale + off;
        cell[d] = (uint32_t)floorf(p);
        w1[d] = p - (float)cell[d];
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

template <typename F>
int dispatch_dc(uint32_t D, uint32_t C, const char* who, F&& f) {
#define ENVIDR_CASE(DD, CC) \
    if (D == DD && C == CC) return f(std::integral_constant<int, DD>{}, std::integral_constant<int, CC>{});
#define ENVIDR_ROW(DD) ENVIDR_CASE(DD, 1) ENVIDR_CASE(DD, 2) ENVIDR_CASE(DD, 4) ENVIDR_CASE(DD, 8)
    ENVIDR_ROW(1) ENVIDR_ROW(2) ENVIDR_ROW(3) ENVIDR_ROW(4) ENVIDR_ROW(5)
#undef ENVIDR_ROW
#undef ENVIDR_CASE
    set_error("%s: unsupported (D=%u, C=%u); D must be 1..5 and C one of 1, 2, 4, 8", who, D, C);
    return ENVIDR_EINVAL;
}

}  // namespace

extern "C" {

int envidr_grid_encode_forward(const float* inputs, const float* embeddings, const int32_t* offsets, float* outputs,
                               uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, float* dy_dx,
                               uint32_t gridtype, int align_corners, envidr_stream_t stream) {
    ENVIDR_REQUIRE(L >= 1 && L <= (uint32_t)kMaxLevels, "grid_encode_forward: L=%u out of range [1,%d]", L, kMaxLevels);
    ENVIDR_REQUIRE(gridtype <= 1, "grid_encode_forward: gridtype must be 0 (hash) or 1 (tiled)");
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(inputs && embeddings && offsets && outputs, "grid_encode_forward: null pointer");
    const LevelScale ls = make_level_scale(L, S, H);
    const uint32_t chunks = ceil_div(B, kBlock);
    const dim3 grid(L * chunks);
    const bool ac = align_corners != 0;
    return dispatch_dc(D, C, "grid_encode_forward", [&](auto d, auto c) {
        constexpr int DD = decltype(d)::value, CC = decltype(c)::value;
        if (dy_dx)
            hipLaunchKernelGGL((k_grid_forward<DD, CC, true>), grid, dim3(kBlock), 0, as_stream(stream), inputs,
                               embeddings, offsets, outputs, B, L, ls, chunks, dy_dx, gridtype, ac);
        else
            hipLaunchKernelGGL((k_grid_forward<DD, CC, false>), grid, dim3(kBlock), 0, as_stream(stream), inputs,
                               embeddings, offsets, outputs, B, L, ls, chunks, dy_dx, gridtype, ac);
        return check_launch("k_grid_forward");
    });
}

int envidr_grid_encode_backward(const float* grad, const float* inputs, const float* embeddings,
                                const int32_t* offsets, float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                uint32_t L, float S, uint32_t H, const float* dy_dx, float* grad_inputs,
                                uint32_t gridtype, int align_corners, envidr_stream_t stream) {
    (void)embeddings;
    ENVIDR_REQUIRE(L >= 1 && L <= (uint32_t)kMaxLevels, "grid_encode_backward: L=%u out of range", L);
    ENVIDR_REQUIRE(gridtype <= 1, "grid_encode_backward: gridtype must be 0 (hash) or 1 (tiled)");
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(grad && inputs && offsets && grad_embeddings, "grid_encode_backward: null pointer");
    ENVIDR_REQUIRE(!dy_dx || grad_inputs, "grid_encode_backward: dy_dx given but grad_inputs is null");
    const LevelScale ls = make_level_scale(L, S, H);
    const uint32_t chunks = ceil_div(B, kBlock);
    const bool ac = align_corners != 0;
    return dispatch_dc(D, C, "grid_encode_backward", [&](auto d, auto c) {
        constexpr int DD = decltype(d)::value, CC = decltype(c)::value;
        hipLaunchKernelGGL((k_grid_backward_table<DD, CC>), dim3(xcd_grid_blocks(L, chunks)), dim3(kBlock), 0,
                           as_stream(stream), grad, inputs, offsets, grad_embeddings, B, L, ls, chunks, gridtype, ac);
        int rc = check_launch("k_grid_backward_table");
        if (rc) return rc;
        if (dy_dx) {
            launch_input_backward<DD, CC>(grad, dy_dx, grad_inputs, B, L, as_stream(stream));
            rc = check_launch("k_grid_input_backward");
        }
        return rc;
    });
}

}  // extern "C"
