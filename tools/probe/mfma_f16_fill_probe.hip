// How should a wave's own vector-ALU work be placed among its v_mfma_f32_32x32x16_f16 (8 passes = 32 cycles, one wave per SIMD)?
//   6 MFMAs per step on two alternating accumulators (the split-precision layer's step); TOTAL vector instructions per step,
//   placed in the gap after every EVERY-th MFMA (EVERY = 1: evenly ... 6: one clump per step).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -o mfma_f16_fill_probe mfma_f16_fill_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int TOTAL, int EVERY, int KIND>
__global__ void __launch_bounds__(64, 1) probe(float* out, unsigned long long* cyc, int iters, float k1, float k2) {
    const unsigned lane = threadIdx.x;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    float x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = (float)(lane + j);
    half8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(lane * 1e-3f); b[j] = (_Float16)(j * 1e-3f); }
    constexpr int PER = TOTAL * EVERY / 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                acc[q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q & 1], 0, 0, 0);
                if (q % EVERY == EVERY - 1) {
#pragma unroll
                    for (int j = 0; j < PER; ++j) {
                        if constexpr (KIND == 0) x[j % 16] = __builtin_fmaf(x[j % 16], k1, k2);
                        else x[j % 16] = __builtin_amdgcn_fmed3f(x[j % 16], k1, k2);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[t][r];
#pragma unroll
    for (int j = 0; j < 16; ++j) sum += x[j];
    out[blockIdx.x * 64 + lane] = sum;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int TOTAL, int EVERY, int KIND = 0>
void run() {
    const int blocks = 1024, iters = 200;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, blocks * 64 * 4); (void)hipMalloc(&cyc, blocks * 8);
    for (int rep = 0; rep < 2; ++rep) {
        probe<TOTAL, EVERY, KIND><<<blocks, 64>>>(out, cyc, iters, 1.0001f, 0.5f);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        if (rep) printf("%2d vector instr per 6 MFMAs, a clump of %2d after every %d MFMA(s): %.2f ticks / MFMA\n", TOTAL, TOTAL * EVERY / 6, EVERY, avg / (iters * 96.0));
    }
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<0, 1>();
    run<6, 1>(); run<6, 2>(); run<6, 3>(); run<6, 6>();
    run<12, 1>(); run<12, 2>(); run<12, 3>(); run<12, 6>();
    run<18, 1>(); run<18, 2>(); run<18, 3>(); run<18, 6>();
    run<24, 1>(); run<24, 2>(); run<24, 3>(); run<24, 6>();
    run<30, 1>(); run<30, 2>(); run<30, 3>(); run<30, 6>();
    run<36, 1>(); run<36, 2>(); run<36, 3>(); run<36, 6>();
    return 0;
}
