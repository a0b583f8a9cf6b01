// Feasibility probe for a 2-waves-per-SIMD design: v_mfma_f32_16x16x4_f32 (8 passes, 32 cycles) hidden layer
// 256 -> 256 on 16 samples (in 16 tiles x 4 regs, out 16 tiles x 4 regs), one 256-B weight fragment per MFMA streamed
// from a 600 KB blob through a register ring, alternating with a block of vector-ALU work that stands for the
// non-matrix sections.  WAVES = waves per SIMD.  Reports SIMD matrix-pipe utilisation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WAVES, int VALU_OPS>
__global__ void __launch_bounds__(64, WAVES) probe(const float* __restrict__ w, float* out, unsigned long long* cyc, int iters, float k1, float k2) {
    const unsigned lane = threadIdx.x;
    f32x4 in[16], acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { in[t][r] = (float)(lane + t + r) * 1e-3f; acc[t][r] = 0; }
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, 1 << 20, 0x00020000);
    constexpr int PF = 32;
    float ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4u, i * 256, 0));
    float x[4] = {1.f + lane, 2.f, 3.f, 4.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // matrix phase: 64 steps x 16 output tiles
#pragma unroll
        for (int kt = 0; kt < 16; ++kt) {
            float bq[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { float y; asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(in[kt][r])); bq[r] = y; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int i = (kt * 4 + r) * 16 + t;
                    const float a = ring[i % PF];
                    ring[i % PF] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4u, ((i + PF) % 2304) * 256, 0));
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq[r], acc[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) { in[t] = acc[t] * 1e-3f; }
        // vector-ALU phase
#pragma unroll 8
        for (int j = 0; j < VALU_OPS / 4; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q] = __builtin_fmaf(x[q], k1, k2);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = x[0] + x[1] + x[2] + x[3];
#pragma unroll
    for (int t = 0; t < 16; ++t) sum += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 64 + lane] = sum;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int WAVES, int VALU_OPS>
void run() {
    const int blocks = 1024 * WAVES, iters = 40;
    float *w, *out; unsigned long long* cyc;
    (void)hipMalloc(&w, 1 << 20); (void)hipMemset(w, 0, 1 << 20);
    (void)hipMalloc(&out, blocks * 64 * 4); (void)hipMalloc(&cyc, blocks * 8);
    for (int rep = 0; rep < 2; ++rep) {
        probe<WAVES, VALU_OPS><<<blocks, 64>>>(w, out, cyc, iters, 1.0001f, 0.5f);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        const double per_iter = avg / iters;                      // ticks one wave needs per iteration
        const double mfma_ticks = 1024.0 * 32.0;                  // matrix-pipe ticks one wave's iteration needs
        if (rep) printf("%d wave(s)/SIMD, %5d VALU ops per 1024 MFMAs: %.0f ticks / iteration / wave -> matrix pipe busy %.1f %%\n",
                        WAVES, VALU_OPS, per_iter, 100.0 * WAVES * mfma_ticks / per_iter);
    }
    (void)hipFree(w); (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<1, 0>(); run<1, 1600>(); run<2, 0>(); run<2, 1600>(); run<2, 3200>();
    return 0;
}
