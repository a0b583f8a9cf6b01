// Do the vector-ALU instructions of one wave overlap the fp32 MFMAs of ANOTHER wave on the same SIMD?
// A workgroup of 8 waves (two per SIMD); by SIMD id (HW_ID[5:4]) and arrival order one wave of each SIMD takes role A, the other role B.
//   mode 0: A = MFMA stream, B idle | 1: A idle, B = VALU stream | 2: A = MFMA, B = VALU | 3: both = alternating MFMA / VALU sections (in phase)
//   4: both alternate, B starts with the other section (anti-phase) | 1N / 2N: modes 2 / 4 with N x `s_nop 15` after every MFMA
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -o cross_wave_probe cross_wave_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void mfma_section(f32x16 (&acc)[4], float a, float b) {
#pragma unroll
    for (int i = 0; i < 224; ++i) { acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 3], 0, 0, 0); }
}
template <int NOPS>
__device__ __forceinline__ void mfma_section_yield(f32x16 (&acc)[4], float a, float b) {
#pragma unroll
    for (int i = 0; i < 224; ++i) {
        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 3], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NOPS; ++q) asm volatile("s_nop 15");
    }
}
__device__ __forceinline__ void valu_section(float (&x)[16], float k1, float k2) {
#pragma unroll
    for (int i = 0; i < 2496; ++i) x[i & 15] = __builtin_fmaf(x[i & 15], k1, k2);
}

template <int MODE>
__global__ void __launch_bounds__(512, 1) probe(float* out, unsigned long long* cyc, int iters, float k1, float k2) {
    __shared__ uint32_t s_cnt[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u;
    const uint32_t simd = __builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));
    uint32_t role = 0;
    if (lane == 0) role = atomicAdd(&s_cnt[simd], 1u);
    role = __builtin_amdgcn_readfirstlane(role);
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    float x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = (float)(lane + j);
    const float a = (float)lane * 1e-3f, b = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { if (role == 0) mfma_section(acc, a, b); }
        if (MODE == 1) { if (role != 0) valu_section(x, k1, k2); }
        if (MODE == 2) { if (role == 0) mfma_section(acc, a, b); else valu_section(x, k1, k2); }
        if (MODE == 3) { mfma_section(acc, a, b); valu_section(x, k1, k2); }
        if (MODE == 4) { if (role == 0) { mfma_section(acc, a, b); valu_section(x, k1, k2); } else { valu_section(x, k1, k2); mfma_section(acc, a, b); } }
        if (MODE >= 10 && MODE < 20) { if (role == 0) mfma_section_yield<MODE - 10>(acc, a, b); else valu_section(x, k1, k2); }
        if (MODE == 32) { if (role == 0) mfma_section(acc, a, b); else { __builtin_amdgcn_s_setprio(3); valu_section(x, k1, k2); } }
        if (MODE == 33) { if (role == 0) { __builtin_amdgcn_s_setprio(3); mfma_section(acc, a, b); } else valu_section(x, k1, k2); }
        if (MODE == 34) {          // anti-phase sections, the VALU section outranks the MFMA section
            if (role == 0) { __builtin_amdgcn_s_setprio(0); mfma_section(acc, a, b); __builtin_amdgcn_s_setprio(3); valu_section(x, k1, k2); }
            else { __builtin_amdgcn_s_setprio(3); valu_section(x, k1, k2); __builtin_amdgcn_s_setprio(0); mfma_section(acc, a, b); }
        }
        if (MODE >= 20 && MODE < 30) { if (role == 0) { mfma_section_yield<MODE - 20>(acc, a, b); valu_section(x, k1, k2); } else { valu_section(x, k1, k2); mfma_section_yield<MODE - 20>(acc, a, b); } }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[t][r];
#pragma unroll
    for (int j = 0; j < 16; ++j) sum += x[j];
    out[blockIdx.x * 512 + threadIdx.x] = sum;
    if (lane == 0) { cyc[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = t1 - t0; cyc[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = simd * 16 + role; }
}

template <int MODE>
void run(const char* what) {
    const int blocks = 256, iters = 50;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, blocks * 512 * 4); (void)hipMalloc(&cyc, blocks * 16 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        probe<MODE><<<blocks, 512>>>(out, cyc, iters, 1.0001f, 0.5f);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 16);
        (void)hipMemcpy(h.data(), cyc, blocks * 16 * 8, hipMemcpyDeviceToHost);
        double avg[2] = {0, 0}; int n[2] = {0, 0}, bad = 0;
        for (int w = 0; w < blocks * 8; ++w) { const int role = (int)(h[2 * w + 1] & 15); if (role > 1) { ++bad; continue; } avg[role] += (double)h[2 * w]; ++n[role]; }
        if (rep) printf("%-70s A %.0f  B %.0f cycles per iteration (224 fp32 MFMAs = 14336 alone; 2496 v_fma = 9984 alone)  [waves with a third wave on their SIMD: %d]\n",
                        what, avg[0] / n[0] / iters, avg[1] / n[1] / iters, bad);
    }
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<0>("A: MFMA stream, B idle");
    run<1>("A idle, B: VALU stream");
    run<2>("A: MFMA stream, B: VALU stream");
    run<3>("both: MFMA section then VALU section, in phase");
    run<4>("both alternate sections, anti-phase");
    run<11>("A: MFMA stream + 1 x s_nop 15 after each, B: VALU stream");
    run<12>("A: MFMA stream + 2 x s_nop 15 after each, B: VALU stream");
    run<13>("A: MFMA stream + 3 x s_nop 15 after each, B: VALU stream");
    run<14>("A: MFMA stream + 4 x s_nop 15 after each, B: VALU stream");
    run<22>("both alternate sections (MFMAs + 2 x s_nop 15), anti-phase");
    run<23>("both alternate sections (MFMAs + 3 x s_nop 15), anti-phase");
    run<32>("A: MFMA stream, B: VALU stream at s_setprio 3");
    run<33>("A: MFMA stream at s_setprio 3, B: VALU stream");
    run<34>("both alternate sections, anti-phase, VALU sections at s_setprio 3");
    return 0;
}
