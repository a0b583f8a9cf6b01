// A wave's own vector-ALU work among its fp32 MFMAs (v_mfma_f32_32x32x2_f32, 16 passes = 64 cycles; one wave per SIMD):
// 224 MFMAs on four rotating accumulators + 11 v_fma per MFMA (the mix of a k_geo_eval32 batch), placed as a clump of 11 * EVERY
// after every EVERY-th MFMA.  Alone the MFMAs take 14.3 k cycles, the 2 464 v_fma 9.9 k.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -o mfma_f32_fill_probe mfma_f32_fill_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int EVERY, int PER_MFMA, bool PK = false>
__global__ void __launch_bounds__(64, 1) probe(float* out, unsigned long long* cyc, int iters, float k1, float k2) {
    const unsigned lane = threadIdx.x;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    float x[16];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 y[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { x[j] = (float)(lane + j); y[j] = f32x2{(float)(lane + j), (float)(lane - j)}; }
    const f32x2 k1v = {k1, k1}, k2v = {k2, k2};
    const float a = (float)lane * 1e-3f, b = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 224; ++i) {
            acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 3], 0, 0, 0);
            if (i % EVERY == EVERY - 1) {
#pragma unroll
                for (int j = 0; j < PER_MFMA * EVERY; ++j) {
                    if constexpr (PK) y[j & 15] = __builtin_elementwise_fma(y[j & 15], k1v, k2v);        // v_pk_fma_f32: two fmas per lane
                    else x[j & 15] = __builtin_fmaf(x[j & 15], k1, k2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[t][r];
#pragma unroll
    for (int j = 0; j < 16; ++j) sum += x[j] + y[j][0] + y[j][1];
    out[blockIdx.x * 64 + lane] = sum;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int EVERY, int PER_MFMA, bool PK = false>
void run() {
    const int blocks = 1024, iters = 50;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, blocks * 64 * 4); (void)hipMalloc(&cyc, blocks * 8);
    for (int rep = 0; rep < 2; ++rep) {
        probe<EVERY, PER_MFMA, PK><<<blocks, 64>>>(out, cyc, iters, 1.0001f, 0.5f);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        if (rep) printf("%2d %s per MFMA, a clump of %4d after every %3d MFMA(s): %.0f cycles per 224 MFMAs + %d v_fma (sum of the two alone: %d)\n", PER_MFMA, PK ? "v_pk_fma" : "v_fma",
                        PER_MFMA * EVERY, EVERY, avg / iters, 224 * PER_MFMA, 14336 + 224 * PER_MFMA * 4);
    }
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<1, 0>();
    run<1, 11>(); run<2, 11>(); run<4, 11>(); run<8, 11>(); run<16, 11>(); run<56, 11>(); run<224, 11>();
    run<1, 4>(); run<4, 4>(); run<16, 4>();
    run<2, 11, true>(); run<4, 11, true>(); run<4, 6, true>(); run<4, 4, true>(); run<4, 2, true>();
    return 0;
}
