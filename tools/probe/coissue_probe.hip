// Which instruction classes overlap fp32 MFMAs (v_mfma_f32_32x32x2_f32, 64 cycles of issue each)?
// Round 4's probes (cross_wave_probe, mfma_f32_fill_probe, valu_overlap_probe) used ONE filler, v_fma_f32 -- the class most likely to share the
// fp32-input MFMA's datapath (MI355X_MICROARCH.md: "f32 in ... at the FP32 vector rate").  k_geo_eval32's 1 718 vector instructions per batch
// are (static count of its ISA): ~520 plain fp32 (mul / add / sub / fma / max / floor), ~333 packed fp32 (v_pk_add / mul / fma), ~560 integer
// (v_add_u32 148, v_sub_u32 65, v_lshl_add_u32 65, v_min_u32 64, v_bitop3 64, v_cndmask 67, v_xor 32, v_mul_lo_u32 26, v_lshl_add_u64 26,
// conversions 48, v_or 19, v_mad_u64_u32 10), 184 v_mov, plus 64 buffer_load_dwordx2 gathers and ~250 LDS instructions.
// This probe prices each class three ways:
//   in-wave      one wave per SIMD: 224 MFMAs with a clump of 4 x PER filler instructions after every 4th MFMA (the best placement round 4 found)
//   same-SIMD    two waves per SIMD: A issues the 224 MFMAs back to back, B the filler stream
//   other-SIMD   one wave per SIMD: the waves on SIMDs 0 and 2 issue MFMAs, those on SIMDs 1 and 3 the filler stream
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -o coissue_probe coissue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Kind { K_FMA = 0, K_INT = 1, K_MOV = 2, K_MULLO = 3, K_CVT = 4, K_PK = 5, K_GATHER = 6, K_LDS = 7, K_INT_NOMUL = 8, K_MIX = 9 };
static const char* kind_name[] = {"v_fma_f32", "integer mix (kernel's)", "v_mov_b32", "v_mul_lo_u32", "v_cvt_f32_u32 / u32_f32", "v_pk_fma_f32",
                                  "buffer_load_dwordx2 (L2-resident gather)", "ds_write_b64 + ds_read_b64", "integer mix without v_mul_lo", "kernel mix (fp32 + pk + int + mov)"};

struct Regs {
    float x[16];
    uint32_t u[16];
    f32x2 y[8];
};

// filler instruction number J (compile time) of kind K; 16 independent dependency chains, every mix has period 16
template <int K, int J>
__device__ __forceinline__ void fill_one(Regs& r, float k1, float k2, uint32_t c1, uint32_t c2, const uint32_t* __restrict__ tab, uint32_t* lds) {
    constexpr int s = J & 15;
    if constexpr (K == K_FMA) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r.x[s]) : "v"(k1), "v"(k2)); }
    else if constexpr (K == K_MOV) { asm volatile("v_mov_b32 %0, %1" : "=v"(r.u[s]) : "v"(r.u[(s + 1) & 15])); }
    else if constexpr (K == K_MULLO) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r.u[s]) : "v"(c1)); }
    else if constexpr (K == K_CVT) {
        if constexpr (J & 1) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(r.x[s]) : "v"(r.u[s]));
        else asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(r.u[s]) : "v"(r.x[s]));
    }
    else if constexpr (K == K_PK) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r.y[s & 7]) : "v"(f32x2{k1, k1}), "v"(f32x2{k2, k2})); }
    else if constexpr (K == K_INT || K == K_INT_NOMUL) {
        // the kernel's integer mix on 16 slots: add 5 (4 + the mul_lo slot), sub 2, lshl_add 2, min 2, xor / bitop 3, cndmask 2; one v_mul_lo_u32 per 16
        // (the kernel: one per 21)
        if constexpr (s < 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r.u[s]) : "v"(c1));
        else if constexpr (s < 6) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r.u[s]) : "v"(c2));
        else if constexpr (s < 8) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r.u[s]) : "v"(c1));
        else if constexpr (s < 10) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r.u[s]) : "v"(c2));
        else if constexpr (s < 13) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r.u[s]) : "v"(c1));
        else if constexpr (s < 15) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r.u[s]) : "v"(c2));
        else if constexpr (K == K_INT) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r.u[s]) : "v"(c1));
        else asm volatile("v_add_u32 %0, %0, %1" : "+v"(r.u[s]) : "v"(c2));
    }
    else if constexpr (K == K_MIX) {
        // proportions of a k_geo_eval32 batch on 16 slots: plain fp32 5, packed fp32 3, integer 6, mov 2
        if constexpr (s < 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r.x[s]) : "v"(k1), "v"(k2));
        else if constexpr (s < 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r.y[s & 7]) : "v"(f32x2{k1, k1}), "v"(f32x2{k2, k2}));
        else if constexpr (s < 11) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r.u[s]) : "v"(c1));
        else if constexpr (s < 13) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r.u[s]) : "v"(c1));
        else if constexpr (s < 14) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r.u[s]) : "v"(c1));
        else asm volatile("v_mov_b32 %0, %1" : "=v"(r.u[s]) : "v"(r.u[(s + 1) & 15]));
    }
    else if constexpr (K == K_GATHER) {
        // a dependent-address 8-byte gather from a 1 MiB table (L2-resident), consumed (waited for) eight fillers later
        const uint32_t idx = (r.u[s] * 2654435761u) & 0x1ffffu;
        const uint2 v = *reinterpret_cast<const uint2*>(tab + 2 * idx);
        r.u[(s + 8) & 15] += v.x + v.y + (uint32_t)J;         // feeds the address of the gather eight fillers on: nothing to hoist
    }
    else if constexpr (K == K_LDS) {
        uint32_t* p = lds + ((threadIdx.x * 2 + (J & 7) * 1024) & 8191);
        if constexpr (J & 1) { const uint2 v = *reinterpret_cast<uint2*>(p); r.u[s] ^= v.x; }
        else *reinterpret_cast<uint2*>(p) = uint2{r.u[s], r.u[(s + 1) & 15]};
    }
}

template <int K, int J0, int... Js>
__device__ __forceinline__ void filler_seq(Regs& r, float k1, float k2, uint32_t c1, uint32_t c2, const uint32_t* __restrict__ tab, uint32_t* lds,
                                           std::integer_sequence<int, Js...>) {
    (fill_one<K, J0 + Js>(r, k1, k2, c1, c2, tab, lds), ...);
}
// N filler instructions (N a compile-time constant: every instruction is selected at compile time, the register arrays stay in registers)
template <int K, int N, int J0 = 0>
__device__ __forceinline__ void filler(Regs& r, float k1, float k2, uint32_t c1, uint32_t c2, const uint32_t* __restrict__ tab, uint32_t* lds) {
    filler_seq<K, J0>(r, k1, k2, c1, c2, tab, lds, std::make_integer_sequence<int, N>{});
}
// in-wave arrangement, step I of 224: one MFMA, and after every 4th a clump of 4 x PER fillers (numbered on through the iteration)
template <int K, int PER, int WHO, int I>
__device__ __forceinline__ void inwave_step(f32x16 (&acc)[4], float a, float b, Regs& r, float k1, float k2, uint32_t c1, uint32_t c2,
                                            const uint32_t* __restrict__ tab, uint32_t* lds) {
    if constexpr (WHO != 2) acc[I & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[I & 3], 0, 0, 0);
    if constexpr (WHO != 1 && (I & 3) == 3) filler<K, 4 * PER, (I - 3) * PER>(r, k1, k2, c1, c2, tab, lds);
    __builtin_amdgcn_sched_barrier(0);
}
template <int K, int PER, int WHO, int... Is>
__device__ __forceinline__ void inwave_iteration(f32x16 (&acc)[4], float a, float b, Regs& r, float k1, float k2, uint32_t c1, uint32_t c2,
                                                 const uint32_t* __restrict__ tab, uint32_t* lds, std::integer_sequence<int, Is...>) {
    (inwave_step<K, PER, WHO, Is>(acc, a, b, r, k1, k2, c1, c2, tab, lds), ...);
}
// the stream of a filler-only wave: NFILL instructions as a run-time loop over clumps of 32 (two periods)
template <int K, int NFILL>
__device__ __forceinline__ void filler_stream(Regs& r, float k1, float k2, uint32_t c1, uint32_t c2, const uint32_t* __restrict__ tab, uint32_t* lds) {
    static_assert(NFILL % 32 == 0);
#pragma unroll 1
    for (int q = 0; q < NFILL / 32; ++q) { filler<K, 32>(r, k1, k2, c1, c2, tab, lds); __builtin_amdgcn_sched_barrier(0); }
}

__device__ __forceinline__ void mfma_section(f32x16 (&acc)[4], float a, float b) {
#pragma unroll
    for (int i = 0; i < 224; ++i) { acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 3], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
}

// ARR 0: in-wave (blockDim 64); 1: same-SIMD pair (blockDim 512: roles by arrival per SIMD); 2: other-SIMD (blockDim 256: role by SIMD parity)
// WHO 0: both streams; 1: only the MFMA stream; 2: only the filler stream
template <int K, int ARR, int PER, int WHO>
__global__ void __launch_bounds__(ARR == 0 ? 64 : ARR == 1 ? 512 : 256, 1)
probe(float* out, unsigned long long* cyc, int iters, float k1, float k2, uint32_t c1, uint32_t c2, const uint32_t* __restrict__ tab) {
    __shared__ uint32_t s_cnt[4];
    __shared__ uint32_t lds[8192 + 64];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u;
    const uint32_t simd = __builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));
    uint32_t role = 0;
    if (ARR == 1) { if (lane == 0) role = atomicAdd(&s_cnt[simd], 1u); role = __builtin_amdgcn_readfirstlane(role); }
    if (ARR == 2) role = simd & 1u;
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0;
    Regs r;
#pragma unroll
    for (int j = 0; j < 16; ++j) { r.x[j] = (float)(lane + j); r.u[j] = lane * 977u + j; }
#pragma unroll
    for (int j = 0; j < 8; ++j) r.y[j] = f32x2{(float)(lane + j), (float)(lane - j)};
    const float a = (float)lane * 1e-3f, b = 0.5f;
    constexpr int NFILL = 224 * PER;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (ARR == 0) {
            inwave_iteration<K, PER, WHO>(acc, a, b, r, k1, k2, c1, c2, tab, lds, std::make_integer_sequence<int, 224>{});
        } else {
            if (role == 0) { if (WHO != 2) mfma_section(acc, a, b); }
            else { if (WHO != 1) filler_stream<K, NFILL>(r, k1, k2, c1, c2, tab, lds); }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) sum += acc[t][q];
#pragma unroll
    for (int j = 0; j < 16; ++j) sum += r.x[j] + (float)r.u[j];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += r.y[j][0] + r.y[j][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    const int waves = blockDim.x >> 6;
    if (lane == 0) { cyc[(blockIdx.x * waves + (threadIdx.x >> 6)) * 2] = t1 - t0; cyc[(blockIdx.x * waves + (threadIdx.x >> 6)) * 2 + 1] = role; }
}

static uint32_t* g_tab = nullptr;

template <int K, int ARR, int PER, int WHO>
void measure(double (&avg)[2]) {
    const int threads = ARR == 0 ? 64 : ARR == 1 ? 512 : 256, waves = threads / 64;
    const int blocks = ARR == 0 ? 1024 : 256, iters = 40;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, (size_t)blocks * threads * 4); (void)hipMalloc(&cyc, (size_t)blocks * waves * 16);
    for (int rep = 0; rep < 2; ++rep) {
        probe<K, ARR, PER, WHO><<<blocks, threads>>>(out, cyc, iters, 1.0001f, 0.5f, 2654435761u, 805459861u, g_tab);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)blocks * waves * 2);
        (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double s[2] = {0, 0}; int n[2] = {0, 0};
        for (int w = 0; w < blocks * waves; ++w) { const int role = (int)(h[2 * w + 1] & 1); s[role] += (double)h[2 * w]; ++n[role]; }
        for (int q = 0; q < 2; ++q) avg[q] = n[q] ? s[q] / n[q] / iters : 0;
    }
    (void)hipFree(out); (void)hipFree(cyc);
}

template <int K, int PER>
void row() {
    double m[2], f[2], both[2];
    // in-wave
    measure<K, 0, PER, 1>(m); measure<K, 0, PER, 2>(f); measure<K, 0, PER, 0>(both);
    const double hidden0 = (m[0] + f[0] - both[0]) / f[0];
    printf("%-42s %2d per MFMA | in-wave: MFMAs %6.0f  fillers %6.0f  together %6.0f  (%3.0f %% of the filler time hidden)", kind_name[K], PER, m[0], f[0], both[0], 100 * hidden0);
    // same SIMD, other wave
    measure<K, 1, PER, 1>(m); measure<K, 1, PER, 2>(f); measure<K, 1, PER, 0>(both);
    printf(" | same SIMD: A %6.0f  B alone %6.0f  B beside A %6.0f (A then %6.0f)", m[0], f[1], both[1], both[0]);
    // other SIMD
    measure<K, 2, PER, 1>(m); measure<K, 2, PER, 2>(f); measure<K, 2, PER, 0>(both);
    printf(" | other SIMD: B alone %6.0f  beside A %6.0f (A %6.0f -> %6.0f)\n", f[1], both[1], m[0], both[0]);
}

int main() {
    (void)hipMalloc(&g_tab, 1 << 20);
    (void)hipMemset(g_tab, 1, 1 << 20);
    printf("cycles per iteration of 224 fp32 MFMAs (14 336 of issue) and 224 x PER filler instructions; k_geo_eval32 has 7.7 vector instructions per MFMA\n");
    row<K_FMA, 8>();
    row<K_INT, 8>();
    row<K_INT_NOMUL, 8>();
    row<K_MOV, 8>();
    row<K_MULLO, 2>();
    row<K_CVT, 4>();
    row<K_PK, 4>();
    row<K_MIX, 8>();
    row<K_GATHER, 1>();
    row<K_LDS, 2>();
    row<K_FMA, 4>();
    row<K_INT, 4>();
    row<K_MIX, 4>();
    return 0;
}
