// issue rate of v_mfma_f32_32x32x16_f16 as a function of the distance between two MFMAs on the same accumulator
// (D accumulators used round-robin by one wave per SIMD), with the A/B operands in VGPRs.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_dep_probe.hip -o mfma_dep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int D>
__global__ void __launch_bounds__(256, 1) k(float* out, int iters, long long* cycles) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 acc[D];
    for (int d = 0; d < D; ++d) for (int r = 0; r < 16; ++r) acc[d][r] = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[d], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int d = 0; d < D; ++d) for (int r = 0; r < 16; ++r) s += acc[d][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}
template <int D> void run(float* out, long long* cyc) {
    const int iters = 20000;
    hipLaunchKernelGGL(k<D>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<D>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("distance %d: %.1f ns per MFMA (%.1f counter ticks), %.0f TFLOP/s\n", D, ms * 1e6 / ((double)iters * D), (double)h / ((double)iters * D),
           1024.0 * iters * D * 32768.0 / (ms * 1e-3) / 1e12);
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    run<1>(out, cyc); run<2>(out, cyc); run<3>(out, cyc); run<4>(out, cyc); run<8>(out, cyc);
    return 0;
}
