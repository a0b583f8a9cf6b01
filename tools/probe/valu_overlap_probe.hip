// Does a wave's own VALU work overlap its own MFMAs?  (one wave per SIMD)
//   NV independent v_fma per MFMA, not feeding the MFMA (mode 0) or feeding its B operand (mode 1)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int FEED>
__global__ void __launch_bounds__(64, 1) probe(float* out, unsigned long long* cyc, int iters, float k1, float k2) {
    const unsigned lane = threadIdx.x;
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (float)(lane + j);
    float a = (float)lane * 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
#pragma unroll
                for (int j = 0; j < NV; ++j) x[j] = __builtin_fmaf(x[j], k1, k2);      // independent chains across j
                const float bv = FEED ? x[0] : a;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[t][r];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += x[j];
    out[blockIdx.x * 64 + lane] = sum;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int FEED>
void run() {
    const int blocks = 256, iters = 200;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, blocks * 64 * 4); (void)hipMalloc(&cyc, blocks * 8);
    for (int rep = 0; rep < 2; ++rep) {
        probe<NV, FEED><<<blocks, 64>>>(out, cyc, iters, 1.0001f, 0.5f);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        if (rep) printf("%d independent v_fma per MFMA, %s: %.2f ticks / MFMA\n", NV, FEED ? "x[0] feeds B" : "not feeding the MFMA", avg / (iters * 128.0));
    }
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<0, 0>(); run<1, 0>(); run<2, 0>(); run<4, 0>(); run<8, 0>();
    run<1, 1>(); run<2, 1>(); run<4, 1>();
    return 0;
}
