// freqencoder operators for gfx950 -- replaces freqencoder/src/freqencoder.cu
// (freq_encode_forward :97, freq_encode_backward :113; declarations freqencoder.h:6-9).
// outputs[b] = [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...] with cos evaluated as sin(. + pi/2)
// in fp32, exactly the reference's formulation.  One lane per output element:
// consecutive lanes write consecutive floats.
#include "common.hip.h"

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

extern "C" {

int envidr_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs,
                               envidr_stream_t stream) {
    ENVIDR_REQUIRE(C == D + 2 * D * deg, "freq_encode_forward: C=%u must equal D + 2*D*deg = %u", C, D + 2 * D * deg);
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(inputs && outputs && D >= 1, "freq_encode_forward: null pointer or D = 0");
    hipLaunchKernelGGL(k_freq_forward, dim3(ceil_div(B * C, kBlock)), dim3(kBlock), 0, as_stream(stream), inputs, B, D, C,
                       outputs);
    return check_launch("k_freq_forward");
}

int envidr_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg,
                                uint32_t C, float* grad_inputs, envidr_stream_t stream) {
    ENVIDR_REQUIRE(C == D + 2 * D * deg, "freq_encode_backward: C=%u must equal D + 2*D*deg", C);
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(grad && outputs && grad_inputs && D >= 1, "freq_encode_backward: null pointer or D = 0");
    hipLaunchKernelGGL(k_freq_backward, dim3(ceil_div(B * D, kBlock)), dim3(kBlock), 0, as_stream(stream), grad, outputs,
                       B, D, deg, C, grad_inputs);
    return check_launch("k_freq_backward");
}

}  // extern "C"
