// checks the operand/result layout assumptions of v_mfma_f32_32x32x16_f16 that mlp_split.hip.h relies on:
//   D[m][n] = sum over (h, i) of A(lane = m + 32 h, slot i) * B(lane = n + 32 h, slot i)
//   D register r of lane (n, h) holds row (r & 3) + 8 (r >> 2) + 4 h
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_f16_probe.hip -o /tmp/mfma_f16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* D) {   // A[64][8], B[64][8] per (lane, slot); D[64][16]
    const int lane = threadIdx.x;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)A[lane * 8 + i]; b[i] = (_Float16)B[lane * 8 + i]; }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[lane * 16 + r] = c[r];
}
int main() {
    float hA[512], hB[512], hD[1024];
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 37 % 17) - 8) / 8.0f; hB[i] = (float)((i * 53 % 13) - 6) / 4.0f; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    double worst = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 16; ++r) {
            const int n = lane & 31, h = lane >> 5, m = (r & 3) + 8 * (r >> 2) + 4 * h;
            double want = 0;
            for (int hh = 0; hh < 2; ++hh)
                for (int i = 0; i < 8; ++i) want += (double)hA[(m + 32 * hh) * 8 + i] * hB[(n + 32 * hh) * 8 + i];
            worst = fmax(worst, fabs(want - hD[lane * 16 + r]));
        }
    printf("mfma_f32_32x32x16_f16 layout check: max |diff| = %g (%s)\n", worst, worst < 1e-3 ? "ok" : "MISMATCH");
    // subnormal fp16 inputs: A = 2^-20 and 3 * 2^-24 (both below fp16's smallest normal 2^-14), B = 1024: flushed -> 0
    for (int i = 0; i < 512; ++i) { hA[i] = 0; hB[i] = 0; }
    for (int m = 0; m < 32; ++m) { hA[m * 8 + 0] = ldexpf(1.0f, -20); hA[m * 8 + 1] = 3 * ldexpf(1.0f, -24); }
    for (int n = 0; n < 32; ++n) { hB[n * 8 + 0] = 1024.0f; hB[n * 8 + 1] = 1024.0f; }
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    const double want_sub = 1024.0 * (ldexp(1.0, -20) + 3 * ldexp(1.0, -24));
    printf("subnormal fp16 operands: D[0][0] = %.9g, exact %.9g (%s)\n", hD[0], want_sub, fabs(hD[0] - want_sub) < 1e-9 ? "honoured" : "FLUSHED");
    return worst < 1e-3 ? 0 : 1;
}
