// freqencoder operators for gfx950 -- replaces freqencoder/src/freqencoder.cu
// (freq_encode_forward :97, freq_encode_backward :113; declarations freqencoder.h:6-9).
// outputs[b] = [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...] with cos evaluated as sin(. + pi/2)
// in fp32, exactly the reference's formulation.  One lane per output element:
// consecutive lanes write consecutive floats.
#include "common.hip.h"
#include "rowio.hip.h"

using namespace envidr;

__global__ void __launch_bounds__(kBlock) k_freq_forward(const float* __restrict__ inputs, uint32_t B, uint32_t D,
                                                         uint32_t C, float* __restrict__ outputs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * C) return;
    const uint32_t b = t / C, c = t - b * C;
    const float* x = inputs + (size_t)b * D;
    if (c < D) {
        outputs[t] = x[c];
    } else {
        const uint32_t col = c / D - 1, d = c % D;
        const float phase = (col & 1u) * (3.141592653589793f / 2);
        outputs[t] = sinf(scalbnf(x[d], (int)(col >> 1)) + phase);
    }
}

// d/dx of the above using the saved outputs: g_x + sum_f 2^f (g_sin * cos - g_cos * sin)
__global__ void __launch_bounds__(kBlock) k_freq_backward(const float* __restrict__ grad,
                                                          const float* __restrict__ outputs, uint32_t B, uint32_t D,
                                                          uint32_t deg, uint32_t C, float* __restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float* g = grad + (size_t)b * C;
    const float* o = outputs + (size_t)b * C;
    float r = g[d];
    g += D; o += D;
    for (uint32_t f = 0; f < deg; ++f) {
        r += scalbnf(1.0f, (int)f) * (g[d] * o[D + d] - g[D + d] * o[d]);
        g += 2 * D; o += 2 * D;
    }
    grad_inputs[t] = r;
}

// D = 3 (every use in the reference's networks): one lane per point, rows of C = 3 + 6 deg floats through LDS (rowio.hip.h) --
// no integer division per output element, 16-byte stores.  Same expressions, element for element, as the generic kernels above.
template <int DEG, bool ALIGNED>
__global__ void __launch_bounds__(64) k_freq_forward3(const float* __restrict__ inputs, uint32_t B, float* __restrict__ outputs) {
    constexpr int D = 3, C = D + 2 * D * DEG;
    __shared__ float s_tile[wave_tile_floats<C>()];
    const uint32_t lane = threadIdx.x, row0 = blockIdx.x * 64u;
    const uint32_t b = min(row0 + lane, B - 1);
    float x[D], v[C];
#pragma unroll
    for (int d = 0; d < D; ++d) { x[d] = inputs[(size_t)b * D + d]; v[d] = x[d]; }
#pragma unroll
    for (int c = D; c < C; ++c) {
        const int col = c / D - 1, d = c % D;
        const float phase = (col & 1) * (3.141592653589793f / 2);
        v[c] = sinf(scalbnf(x[d], col >> 1) + phase);
    }
    wave_store_rows<C, ALIGNED>(s_tile, v, outputs + (size_t)row0 * C, min(64u, B - row0), lane);
}

template <int DEG, bool ALIGNED>
__global__ void __launch_bounds__(64) k_freq_backward3(const float* __restrict__ grad, const float* __restrict__ outputs, uint32_t B,
                                                       float* __restrict__ grad_inputs) {
    constexpr int D = 3, C = D + 2 * D * DEG;
    __shared__ float s_tile[wave_tile_floats<C>()];
    const uint32_t lane = threadIdx.x, row0 = blockIdx.x * 64u, rows = min(64u, B - row0);
    float g[C], o[C];
    wave_load_rows<C, ALIGNED>(s_tile, g, grad + (size_t)row0 * C, rows, lane);
    wave_load_rows<C, ALIGNED>(s_tile, o, outputs + (size_t)row0 * C, rows, lane);
    float r[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        r[d] = g[d];
#pragma unroll
        for (int f = 0; f < DEG; ++f) {
            const int at = D + 2 * D * f;
            r[d] += scalbnf(1.0f, f) * (g[at + d] * o[at + D + d] - g[at + D + d] * o[at + d]);
        }
    }
    wave_store_rows<D, ALIGNED>(s_tile, r, grad_inputs + (size_t)row0 * D, rows, lane);
}

extern "C" {

int envidr_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs,
                               envidr_stream_t stream) {
    ENVIDR_REQUIRE(C == D + 2 * D * deg, "freq_encode_forward: C=%u must equal D + 2*D*deg = %u", C, D + 2 * D * deg);
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(inputs && outputs && D >= 1, "freq_encode_forward: null pointer or D = 0");
    if (D == 3 && deg >= 1 && deg <= 10) {
        const dim3 grid(ceil_div(B, 64)), block(64);
        hipStream_t s = as_stream(stream);
        const bool al = aligned16(outputs);
#define ENVIDR_FREQ(DEG)                                                                                    \
    case DEG:                                                                                               \
        if (al) hipLaunchKernelGGL((k_freq_forward3<DEG, true>), grid, block, 0, s, inputs, B, outputs);      \
        else hipLaunchKernelGGL((k_freq_forward3<DEG, false>), grid, block, 0, s, inputs, B, outputs);        \
        break;
        switch (deg) { ENVIDR_FREQ(1) ENVIDR_FREQ(2) ENVIDR_FREQ(3) ENVIDR_FREQ(4) ENVIDR_FREQ(5) ENVIDR_FREQ(6) ENVIDR_FREQ(7) ENVIDR_FREQ(8) ENVIDR_FREQ(9) ENVIDR_FREQ(10) }
#undef ENVIDR_FREQ
        return check_launch("k_freq_forward3");
    }
    hipLaunchKernelGGL(k_freq_forward, dim3(ceil_div(B * C, kBlock)), dim3(kBlock), 0, as_stream(stream), inputs, B, D, C,
                       outputs);
    return check_launch("k_freq_forward");
}

int envidr_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg,
                                uint32_t C, float* grad_inputs, envidr_stream_t stream) {
    ENVIDR_REQUIRE(C == D + 2 * D * deg, "freq_encode_backward: C=%u must equal D + 2*D*deg", C);
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(grad && outputs && grad_inputs && D >= 1, "freq_encode_backward: null pointer or D = 0");
    if (D == 3 && deg >= 1 && deg <= 10) {
        const dim3 grid(ceil_div(B, 64)), block(64);
        hipStream_t s = as_stream(stream);
        const bool al = aligned16(grad) && aligned16(outputs) && aligned16(grad_inputs);
#define ENVIDR_FREQ(DEG)                                                                                                  \
    case DEG:                                                                                                             \
        if (al) hipLaunchKernelGGL((k_freq_backward3<DEG, true>), grid, block, 0, s, grad, outputs, B, grad_inputs);        \
        else hipLaunchKernelGGL((k_freq_backward3<DEG, false>), grid, block, 0, s, grad, outputs, B, grad_inputs);          \
        break;
        switch (deg) { ENVIDR_FREQ(1) ENVIDR_FREQ(2) ENVIDR_FREQ(3) ENVIDR_FREQ(4) ENVIDR_FREQ(5) ENVIDR_FREQ(6) ENVIDR_FREQ(7) ENVIDR_FREQ(8) ENVIDR_FREQ(9) ENVIDR_FREQ(10) }
#undef ENVIDR_FREQ
        return check_launch("k_freq_backward3");
    }
    hipLaunchKernelGGL(k_freq_backward, dim3(ceil_div(B * D, kBlock)), dim3(kBlock), 0, as_stream(stream), grad, outputs,
                       B, D, deg, C, grad_inputs);
    return check_launch("k_freq_backward");
}

}  // extern "C"
