// shencoder operators for gfx950 -- replaces shencoder/src/shencoder.cu (sh_encode_forward :404,
// sh_encode_backward :421; declarations shencoder.h:9-10).  One lane per direction; the basis is
// evaluated from generated coefficient tables (sh_core.hip.h) rather than a hard-coded expression
// list, in per-lane registers, then written as contiguous rows.
#include "sh_core.hip.h"
#include "rowio.hip.h"

using namespace envidr;

// One lane per direction, one wave per workgroup; output rows (and the three gradient blocks of a dy_dx row) leave through
// LDS as 16-byte stores (rowio.hip.h).
template <int DEG, bool GRAD, bool ALIGNED>
__global__ void __launch_bounds__(64) k_sh_forward(const float* __restrict__ inputs, float* __restrict__ outputs, uint32_t B,
                                                   float* __restrict__ dy_dx) {
    constexpr int C2 = DEG * DEG;
    __shared__ float s_tile[wave_tile_floats<(GRAD ? 3 : 1) * C2>()];
    const uint32_t lane = threadIdx.x, row0 = blockIdx.x * 64u, rows = min(64u, B - row0);
    const uint32_t b = min(row0 + lane, B - 1);
    const float x = inputs[3 * (size_t)b], y = inputs[3 * (size_t)b + 1], z = inputs[3 * (size_t)b + 2];
    float o[C2], g[GRAD ? 3 * C2 : 1];
    sh_eval<DEG, GRAD>(x, y, z, o, g, g + (GRAD ? C2 : 0), g + (GRAD ? 2 * C2 : 0));
    wave_store_rows<C2, ALIGNED>(s_tile, o, outputs + (size_t)row0 * C2, rows, lane);
    if constexpr (GRAD) wave_store_rows<3 * C2, ALIGNED>(s_tile, g, dy_dx + (size_t)row0 * 3 * C2, rows, lane);
}

// grad_inputs[b,d] += sum_ch grad[b,ch] * dy_dx[b,d,ch]   (accumulates, like the reference; same summation order).
// One lane per (point, dimension); rows of 4 k floats are read 16 bytes at a time.
template <bool VEC>
__global__ void __launch_bounds__(kBlock) k_sh_backward(const float* __restrict__ grad, uint32_t B, uint32_t D, uint32_t C2,
                                                        const float* __restrict__ dy_dx, float* __restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float* g = grad + (size_t)b * C2;
    const float* j = dy_dx + ((size_t)b * D + d) * C2;
    float acc = grad_inputs[t];
    if constexpr (VEC) {
        for (uint32_t ch = 0; ch < C2; ch += 4) {
            const float4 gv = *reinterpret_cast<const float4*>(g + ch), jv = *reinterpret_cast<const float4*>(j + ch);
            acc += gv.x * jv.x; acc += gv.y * jv.y; acc += gv.z * jv.z; acc += gv.w * jv.w;
        }
    } else {
        for (uint32_t ch = 0; ch < C2; ++ch) acc += g[ch] * j[ch];
    }
    grad_inputs[t] = acc;
}

// ---- the at::Half side of the reference's dispatch (shencoder.cu:413,435) -------------------------------------------------------
// Forward: half inputs / outputs / dy_dx, the basis evaluated in fp32 from the widened direction and rounded ONCE.  The
// reference's at::Half instantiation rounds every monomial it forms (x2 = H(x*x), x4 = H(x2*x2), xyz = H(H(x*y)*z) ...) before
// its fp32 polynomial expressions use them, so it sits a few fp16 ulp away from the exact basis; this kernel is the exact basis
// to fp16 rounding, i.e. it agrees with the reference's half kernel to within those few ulp (bound measured against the
// reference's own template in tests/test_oracle_pinning.py), not bit for bit -- that would take its expression list term by term.
// (The reference's Python never reaches either: sphere_harmonics.py:16 casts to float32 "for better precision".)
// Backward: Half product, Half sum per term, exactly the reference's `grad_inputs[t] += grad[ch] * dy_dx[ch]`.
typedef _Float16 h16;
template <int DEG, bool GRAD>
__global__ void __launch_bounds__(kBlock) k_sh_forward_h(const h16* __restrict__ inputs, h16* __restrict__ outputs, uint32_t B,
                                                         h16* __restrict__ dy_dx) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    constexpr int C2 = DEG * DEG;
    const float x = (float)inputs[3 * (size_t)b], y = (float)inputs[3 * (size_t)b + 1], z = (float)inputs[3 * (size_t)b + 2];
    float o[C2], gx[GRAD ? C2 : 1], gy[GRAD ? C2 : 1], gz[GRAD ? C2 : 1];
    sh_eval<DEG, GRAD>(x, y, z, o, gx, gy, gz);
    h16* po = outputs + (size_t)b * C2;
#pragma unroll
    for (int i = 0; i < C2; ++i) po[i] = (h16)o[i];
    if constexpr (GRAD) {
        h16* pg = dy_dx + (size_t)b * 3 * C2;
#pragma unroll
        for (int i = 0; i < C2; ++i) { pg[i] = (h16)gx[i]; pg[C2 + i] = (h16)gy[i]; pg[2 * C2 + i] = (h16)gz[i]; }
    }
}

__global__ void __launch_bounds__(kBlock) k_sh_backward_h(const h16* __restrict__ grad, uint32_t B, uint32_t D, uint32_t C2,
                                                          const h16* __restrict__ dy_dx, h16* __restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const h16* g = grad + (size_t)b * C2;
    const h16* j = dy_dx + ((size_t)b * D + d) * C2;
    h16 acc = grad_inputs[t];
    for (uint32_t ch = 0; ch < C2; ++ch) acc = (h16)((float)acc + (float)(h16)((float)g[ch] * (float)j[ch]));
    grad_inputs[t] = acc;
}

extern "C" {

int envidr_sh_encode_forward_f16(const uint16_t* inputs, uint16_t* outputs, uint32_t B, uint32_t D, uint32_t C, uint16_t* dy_dx,
                                 envidr_stream_t stream) {
    ENVIDR_REQUIRE(D == 3, "sh_encode_forward_f16: input dim must be 3 (got %u)", D);
    ENVIDR_REQUIRE(C >= 1 && C <= 8, "sh_encode_forward_f16: degree must be in [1, 8] (got %u)", C);
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(inputs && outputs, "sh_encode_forward_f16: null pointer");
    const dim3 grid(ceil_div(B, kBlock)), block(kBlock);
    hipStream_t s = as_stream(stream);
    const h16* in = reinterpret_cast<const h16*>(inputs);
    h16* out = reinterpret_cast<h16*>(outputs);
    h16* dy = reinterpret_cast<h16*>(dy_dx);
#define ENVIDR_SH(DEG)                                                                                   \
    case DEG:                                                                                            \
        if (dy) hipLaunchKernelGGL((k_sh_forward_h<DEG, true>), grid, block, 0, s, in, out, B, dy);       \
        else hipLaunchKernelGGL((k_sh_forward_h<DEG, false>), grid, block, 0, s, in, out, B, dy);         \
        break;
    switch (C) { ENVIDR_SH(1) ENVIDR_SH(2) ENVIDR_SH(3) ENVIDR_SH(4) ENVIDR_SH(5) ENVIDR_SH(6) ENVIDR_SH(7) ENVIDR_SH(8) }
#undef ENVIDR_SH
    return check_launch("k_sh_forward_h");
}

int envidr_sh_encode_backward_f16(const uint16_t* grad, const uint16_t* inputs, uint32_t B, uint32_t D, uint32_t C,
                                  const uint16_t* dy_dx, uint16_t* grad_inputs, envidr_stream_t stream) {
    (void)inputs;
    ENVIDR_REQUIRE(D == 3, "sh_encode_backward_f16: input dim must be 3 (got %u)", D);
    ENVIDR_REQUIRE(C >= 1 && C <= 8, "sh_encode_backward_f16: degree must be in [1, 8] (got %u)", C);
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(grad && dy_dx && grad_inputs, "sh_encode_backward_f16: null pointer");
    hipLaunchKernelGGL(k_sh_backward_h, dim3(ceil_div(B * D, kBlock)), dim3(kBlock), 0, as_stream(stream), reinterpret_cast<const h16*>(grad), B,
                       D, C * C, reinterpret_cast<const h16*>(dy_dx), reinterpret_cast<h16*>(grad_inputs));
    return check_launch("k_sh_backward_h");
}

int envidr_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C, float* dy_dx,
                             envidr_stream_t stream) {
    ENVIDR_REQUIRE(D == 3, "sh_encode_forward: input dim must be 3 (got %u)", D);
    ENVIDR_REQUIRE(C >= 1 && C <= 8, "sh_encode_forward: degree must be in [1, 8] (got %u)", C);
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(inputs && outputs, "sh_encode_forward: null pointer");
    const dim3 grid(ceil_div(B, 64)), block(64);
    hipStream_t s = as_stream(stream);
    const bool al = aligned16(outputs) && (!dy_dx || aligned16(dy_dx));
#define ENVIDR_SH(DEG)                                                                                                  \
    case DEG:                                                                                                           \
        if (dy_dx && al) hipLaunchKernelGGL((k_sh_forward<DEG, true, true>), grid, block, 0, s, inputs, outputs, B, dy_dx);  \
        else if (dy_dx) hipLaunchKernelGGL((k_sh_forward<DEG, true, false>), grid, block, 0, s, inputs, outputs, B, dy_dx);  \
        else if (al) hipLaunchKernelGGL((k_sh_forward<DEG, false, true>), grid, block, 0, s, inputs, outputs, B, dy_dx);     \
        else hipLaunchKernelGGL((k_sh_forward<DEG, false, false>), grid, block, 0, s, inputs, outputs, B, dy_dx);            \
        break;
    switch (C) { ENVIDR_SH(1) ENVIDR_SH(2) ENVIDR_SH(3) ENVIDR_SH(4) ENVIDR_SH(5) ENVIDR_SH(6) ENVIDR_SH(7) ENVIDR_SH(8) }
#undef ENVIDR_SH
    return check_launch("k_sh_forward");
}

int envidr_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C,
                              const float* dy_dx, float* grad_inputs, envidr_stream_t stream) {
    (void)inputs;
    ENVIDR_REQUIRE(D == 3, "sh_encode_backward: input dim must be 3 (got %u)", D);
    ENVIDR_REQUIRE(C >= 1 && C <= 8, "sh_encode_backward: degree must be in [1, 8] (got %u)", C);
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(grad && dy_dx && grad_inputs, "sh_encode_backward: null pointer");
    if ((C * C) % 4 == 0 && aligned16(grad) && aligned16(dy_dx))
        hipLaunchKernelGGL(k_sh_backward<true>, dim3(ceil_div(B * D, kBlock)), dim3(kBlock), 0, as_stream(stream), grad, B, D, C * C, dy_dx, grad_inputs);
    else
        hipLaunchKernelGGL(k_sh_backward<false>, dim3(ceil_div(B * D, kBlock)), dim3(kBlock), 0, as_stream(stream), grad, B, D, C * C, dy_dx, grad_inputs);
    return check_launch("k_sh_backward");
}

}  // extern "C"
