// grad_inputs[b,d] = sum_{l,c} grad[l,b,c] * dy_dx[b,l,d,c]   (hashencoder.cu:346-372, gridencoder.cu:308-335: the same kernel in both
// extensions), with the rows moved through LDS.  Shared by hashencoder.hip and gridencoder.hip.
#pragma once
#include "grid_core.hip.h"

namespace envidr {

// e / d for e < 128 * 127 as one multiply-high: M = floor(2^32 / d) + 1 is exact while e (M d - 2^32) < 2^32, i.e. for e < 2^32 / d
inline uint32_t row_division_magic(uint32_t d) { return (uint32_t)((1ull << 32) / d) + 1u; }

// A point's row is L D C contiguous floats.  With one lane per (point, dimension) reading global memory directly (the reference's form, kept
// as k_input_backward_long for rows too long for LDS) the D lanes of a point read interleaved 4 C-byte pieces of it --
// every load instruction touches 64 pieces spread over 8 KiB.  Here ONE WAVE
// copies the rows of 64 points (one contiguous range, 16-byte loads, eight in flight per lane before the first LDS store) and their
// gradient rows (contiguous per level) into LDS, then each lane walks its own point's row there (odd pitch: conflict-free): same order
// of additions per (point, dimension) -- l outer, c inner -- same bits.  One wave per workgroup: no barrier couples the waves of a CU, so
// while one computes the others' loads are in flight.  7.7 M points: 1.27 ms (kernel above) -> 0.95 ms (rows of 128 points per 256-thread
// workgroup, round 5) -> this form, tools/ops_bench.py; the reference's kernel compiled for this GPU: 0.87 ms.
constexpr uint32_t kWaveRows = 64;
template <int D, int C>
__global__ void __launch_bounds__(64) k_input_backward_rows(const float* __restrict__ grad, const float* __restrict__ dy_dx,
                                                           float* __restrict__ grad_inputs, uint32_t B, uint32_t L, uint32_t row_magic) {
    extern __shared__ float s_jrows[];                      // [64][pitch] dy_dx rows, then [L][64][C] gradient rows, one spare slot
    const uint32_t row_floats = L * D * C, pitch = row_floats | 1u, pad = pitch - row_floats;
    float* s_grad = s_jrows + kWaveRows * pitch;
    const uint32_t spare = kWaveRows * (pitch + L * C);
    const uint32_t lane = threadIdx.x;
    const uint32_t b0 = blockIdx.x * kWaveRows, points = min(kWaveRows, B - b0);
    const float* src = dy_dx + (size_t)b0 * row_floats;
    const uint32_t total = points * row_floats;
    // every load of the wave is issued before the first LDS store (the common row, 96 floats x 64 points, is 24 + 16 loads per lane)
    constexpr uint32_t U = 24, UG = 16;
    const uint32_t pc = min(lane, points - 1);
    Feat<C> g[UG];
#pragma unroll
    for (uint32_t u = 0; u < UG; ++u) g[u] = load_row<C>(grad + (size_t)min(u, L - 1) * B * C, b0 + pc);
    uint32_t done = 0;
    if ((reinterpret_cast<uintptr_t>(src) & 15u) == 0 && total >= 4) {
        const uint32_t vec_end = total & ~3u, last = vec_end - 4;
        auto stage = [&](uint32_t base) {
            float4 v[U];
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float4*>(src + min(base + (u * 64 + lane) * 4, last));
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                // (no branch around the stores: the compiler would sink the load into it and drain every outstanding load there;
                //  lanes past the end write a spare slot behind the gradient rows)
                const uint32_t e = base + (u * 64 + lane) * 4;
                const float q[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) s_jrows[e < vec_end ? (e + k) + __umulhi(e + k, row_magic) * pad : spare] = q[k];
            }
        };
        stage(0);           // (straight-line: a loop header would make the compiler drain the gradient loads first)
        for (uint32_t base = 64 * 4 * U; base < vec_end; base += 64 * 4 * U) stage(base);
        done = vec_end;
    }
    for (uint32_t e = done + lane; e < total; e += 64) s_jrows[e + __umulhi(e, row_magic) * pad] = src[e];
    // gradient rows: level l's values of these points are points * C contiguous floats
#pragma unroll
    for (uint32_t u = 0; u < UG; ++u)
        if (u < L) {
#pragma unroll
            for (int c = 0; c < C; ++c) s_grad[(u * kWaveRows + lane) * C + c] = g[u].v[c];
        }
    for (uint32_t l = UG; l < L; ++l) {
        const Feat<C> gl = load_row<C>(grad + (size_t)l * B * C, b0 + pc);
#pragma unroll
        for (int c = 0; c < C; ++c) s_grad[(l * kWaveRows + lane) * C + c] = gl.v[c];
    }
    __syncthreads();
    if (lane >= points) return;
    const float* j = s_jrows + lane * pitch;
    float acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0;
    for (uint32_t l = 0; l < L; ++l) {
        float g[C];
#pragma unroll
        for (int c = 0; c < C; ++c) g[c] = s_grad[(l * kWaveRows + lane) * C + c];
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[d] += g[c] * j[l * D * C + d * C + c];
    }
#pragma unroll
    for (int d = 0; d < D; ++d) grad_inputs[(size_t)(b0 + lane) * D + d] = acc[d];
}

inline bool input_rows_fit(uint32_t L, uint32_t D, uint32_t C) { return L * D * C <= 127; }          // <= 64 KiB of LDS, magic division valid

// Rows too long for LDS (L D C > 127 floats: e.g. 24 levels of 3 x 2): one lane per (point, dimension) reading global memory, the reference's
// form; D and C are run-time values here -- ONE kernel instead of an instantiation per (D, C) of both extensions for a path no shipped
// configuration takes.  Same order of additions (l outer, c inner): same bits.
static __global__ void __launch_bounds__(kBlock) k_input_backward_long(const float* __restrict__ grad, const float* __restrict__ dy_dx, float* __restrict__ grad_inputs,
                                                               uint32_t B, uint32_t L, uint32_t D, uint32_t C) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float* j = dy_dx + (size_t)b * L * D * C + d * C;
    float acc = 0;
    for (uint32_t l = 0; l < L; ++l)
        for (uint32_t c = 0; c < C; ++c) acc += grad[((size_t)l * B + b) * C + c] * j[(size_t)l * D * C + c];
    grad_inputs[t] = acc;
}

template <int D, int C>
inline void launch_input_backward_rows(const float* grad, const float* dy_dx, float* grad_inputs, uint32_t B, uint32_t L, hipStream_t stream) {
    const uint32_t row_floats = L * D * C;
    hipLaunchKernelGGL((k_input_backward_rows<D, C>), dim3(ceil_div(B, kWaveRows)), dim3(64), (kWaveRows * ((row_floats | 1u) + L * C) + 1) * sizeof(float),
                       stream, grad, dy_dx, grad_inputs, B, L, row_division_magic(row_floats));
}

// grad_inputs of a (D, C) grid operator, whichever form the row length allows
template <int D, int C>
inline void launch_input_backward(const float* grad, const float* dy_dx, float* grad_inputs, uint32_t B, uint32_t L, hipStream_t stream) {
    if (input_rows_fit(L, D, C)) launch_input_backward_rows<D, C>(grad, dy_dx, grad_inputs, B, L, stream);
    else hipLaunchKernelGGL(k_input_backward_long, dim3(ceil_div(B * (uint32_t)D, kBlock)), dim3(kBlock), 0, stream, grad, dy_dx, grad_inputs, B, L, (uint32_t)D, (uint32_t)C);
}

}  // namespace envidr
