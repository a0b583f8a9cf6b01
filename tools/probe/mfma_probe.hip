// Microbenchmark: cycles per v_mfma_f32_32x32x2_f32 under the instruction mixes the fused kernel uses.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(64, 1) probe(const float* __restrict__ w, float* out, unsigned long long* cyc, int iters) {
    const unsigned lane = threadIdx.x;
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    float b[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) b[r] = (float)(lane + r) * 1e-3f;
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, 1 << 20, 0x00020000);
    float ring[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) ring[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4u, i * 256, 0));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int i = s * 8 + t;
                float a;
                if constexpr (MODE == 0) a = ring[i % 32];                                   // MFMA only, A from a fixed register
                else {
                    a = ring[i % 32];                                                         // ring take + refill
                    ring[i % 32] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4u, ((i + 32) % 128) * 256, 0));
                }
                float bv = b[s];
                if constexpr (MODE == 2) bv = fmaxf(bv, 0.0f) ;                               // ReLU at operand fetch
                if constexpr (MODE == 3) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[0], 0, 0, 0);   // one dependent chain
                else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[t], 0, 0, 0);
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
    out[blockIdx.x * 64 + lane] = sum;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

// the same per-MFMA work as MODE 2, but as straight-line code far larger than the instruction cache
template <int STEPS>
__global__ void __launch_bounds__(64, 1) probe_big(const float* __restrict__ w, float* out, unsigned long long* cyc, int iters) {
    const unsigned lane = threadIdx.x;
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    float b[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) b[r] = (float)(lane + r) * 1e-3f;
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, 1 << 20, 0x00020000);
    float ring[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) ring[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4u, i * 256, 0));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int i = s * 8 + t;
                const float a = ring[i % 32];
                ring[i % 32] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4u, ((i + 32) % 3072) * 256, 0));
                const float bv = fmaxf(b[s % 16], 0.0f);
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
    out[blockIdx.x * 64 + lane] = sum;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int STEPS>
void run_big(int blocks) {
    float *w, *out; unsigned long long* cyc;
    hipMalloc(&w, 1 << 20); hipMemset(w, 0, 1 << 20);
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
    const int iters = 51200 / (STEPS * 8) * 4;
    for (int rep = 0; rep < 2; ++rep) {
        probe_big<STEPS><<<blocks, 64>>>(w, out, cyc, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        if (rep) printf("straight-line body of %5d MFMAs (~%4d KB of code) blocks %4d: %.2f ticks / MFMA\n", STEPS * 8, STEPS * 8 * 28 / 1024, blocks, avg / ((double)iters * STEPS * 8));
    }
    hipFree(w); hipFree(out); hipFree(cyc);
}

template <int MODE>
void run(const char* name, int blocks) {
    float *w, *out; unsigned long long* cyc;
    hipMalloc(&w, 1 << 20); hipMemset(w, 0, 1 << 20);
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
    const int iters = 400;
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        probe<MODE><<<blocks, 64>>>(w, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        const double n = (double)iters * 128;
        if (rep) printf("%-44s blocks %4d: %.2f s_memtime ticks / MFMA, %.2f ns / MFMA (wall), => tick = %.3f ns\n", name, blocks, avg / n, ms * 1e6 / n, ms * 1e6 / avg);
    }
    hipFree(w); hipFree(out); hipFree(cyc);
}

int main() {
    for (int blocks : {4, 1024}) {
        run<0>("mfma only, 8 accumulators", blocks);
        run<1>("+ ring take/refill (buffer_load per mfma)", blocks);
        run<2>("+ v_max on the B operand", blocks);
        run<3>("ring + single dependent accumulator", blocks);
    }
    for (int blocks : {4, 1024}) {
        run_big<16>(blocks);
        run_big<128>(blocks);
        run_big<320>(blocks);
        run_big<1280>(blocks);
    }
    return 0;
}
