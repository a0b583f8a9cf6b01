// What does a returnless fp32 LDS atomic cost on gfx950, as a function of the lanes taking part and of the address pattern?
// 256 workgroups x 1024 threads (16 waves per CU, like k_table_scatter_lds), 128 KiB of accumulators, every wave issues
// `iters` x 8 atomics; cycles per atomic instruction and CU = elapsed / (iters * 8 * 16 waves).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -o lds_atomic_probe lds_atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

template <int MODE>
__global__ void __launch_bounds__(1024) probe(float* out, unsigned long long* cyc, int iters, uint32_t active, uint32_t seed) {
    __shared__ float s_acc[32768];
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) s_acc[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t h = (threadIdx.x + blockIdx.x * 1024u) * 2654435761u + seed;
    const bool on = lane < active;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            h = h * 1664525u + 1013904223u;
            uint32_t at;
            if (MODE == 0 || MODE == 5) at = (h >> 9) & 32767u;                        // random rows
            else if (MODE == 1) at = ((h >> 9) & 32767u & ~63u) | lane;   // conflict-free: lane = bank
            else if (MODE == 2) at = ((h >> 9) & 511u) * 64u;             // all lanes on one bank, different rows
            else at = (h >> 20) & 32767u & ~0u & (32767u ^ 63u);          // wave-varying, same address in every lane? no: per lane random multiple of 64
            if (on) {
                if (MODE == 4) s_acc[at] = (float)k;                      // plain store for comparison
                else if (MODE == 5) {                                       // the scatter kernel's pair: random values, two channel arrays
                    const float v = __uint_as_float(0x3f800000u | (h & 0x7fffffu)) - 1.5f;
                    __hip_atomic_fetch_add(&s_acc[at & 16383u], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&s_acc[16384u + (at & 16383u)], -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                else __hip_atomic_fetch_add(&s_acc[at], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    float sum = 0;
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) sum += s_acc[i];
    out[blockIdx.x * 1024 + threadIdx.x] = sum;
    if (lane == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* what, uint32_t active) {
    const int blocks = 256, iters = 2000;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, blocks * 1024 * 4); (void)hipMalloc(&cyc, blocks * 16 * 8);
    double avg = 0;
    for (int rep = 0; rep < 2; ++rep) {
        probe<MODE><<<blocks, 1024>>>(out, cyc, iters, active, 12345u + rep);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 16);
        (void)hipMemcpy(h.data(), cyc, blocks * 16 * 8, hipMemcpyDeviceToHost);
        avg = 0;
        for (auto v : h) avg += (double)v;
        avg /= h.size();
    }
    // s_memtime ticks at 100 MHz on this chip: report ticks and the per-instruction share of a CU
    printf("%-50s active lanes %2u: %.1f memtime ticks per wave; per atomic instruction and CU: %.3f ticks\n", what, active, avg, avg / (iters * 8.0 * 16.0));
    (void)hipFree(out); (void)hipFree(cyc);
}
// half of the waves (two on every SIMD) run a VALU stream (or plain LDS writes), the others LDS atomics: does either slow the other down?
template <int OTHER>    // 0: VALU stream, 1: sparse plain ds_write
__global__ void __launch_bounds__(1024) mix(float* out, unsigned long long* cyc, int iters, int do_atomics, int do_other, uint32_t seed) {
    __shared__ float s_acc[32768];
    __shared__ float s_q[16][64];
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) s_acc[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t h = (threadIdx.x + blockIdx.x * 1024u) * 2654435761u + seed;
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = (float)(lane + j);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if ((wave >> 2) & 1) {          // (waves go to SIMDs round-robin: every SIMD gets two waves of each kind)
        if (do_other) {
            for (int it = 0; it < iters; ++it) {
                if (OTHER == 0) {
#pragma unroll
                    for (int k = 0; k < 240; ++k) x[k & 7] = __builtin_fmaf(x[k & 7], 1.0001f, 0.5f);      // ~960 cycles of VALU issue
                } else if (OTHER == 1) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) { if (lane < 14) s_q[wave][(lane + k) & 63] = x[k]; x[k] += 1.0f; }
                } else {
                    // 8 gathers of 12-byte-strided floats out of a 92 MB array (like the scatter kernel's point loads), then use them
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { h = h * 1664525u + 1013904223u; v[k] = out[(h >> 8) % (256u * 1024u)]; }
#pragma unroll
                    for (int k = 0; k < 8; ++k) x[k] += v[k];
                }
            }
        }
    } else if (do_atomics) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                h = h * 1664525u + 1013904223u;
                __hip_atomic_fetch_add(&s_acc[(h >> 9) & 32767u], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    float sum = 0;
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) sum += s_acc[i];
    for (int j = 0; j < 8; ++j) sum += x[j];
    out[blockIdx.x * 1024 + threadIdx.x] = sum + s_q[wave][lane];
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}
// every wave alternates a VALU section and an atomic section (what the waves of k_table_scatter_lds do): does the workgroup take
// max(VALU, LDS) or their sum?  stagger: wave w sleeps w * stagger * 64 cycles first
__global__ void __launch_bounds__(1024) alternate(float* out, unsigned long long* cyc, int iters, int valu_steps, int atomics, int stagger, uint32_t seed) {
    __shared__ float s_acc[32768];
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) s_acc[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t h = (threadIdx.x + blockIdx.x * 1024u) * 2654435761u + seed;
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = (float)(lane + j);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t w = 0; w < wave * (uint32_t)stagger; ++w) __builtin_amdgcn_s_sleep(1);
    for (int it = 0; it < iters; ++it) {
        for (int v = 0; v < valu_steps; ++v) {
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k & 7] = __builtin_fmaf(x[k & 7], 1.0001f, 0.5f);
        }
        for (int a = 0; a < atomics; ++a) {
            h = h * 1664525u + 1013904223u;
            __hip_atomic_fetch_add(&s_acc[(h >> 9) & 32767u], x[a & 7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    float sum = 0;
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) sum += s_acc[i];
    for (int j = 0; j < 8; ++j) sum += x[j];
    out[blockIdx.x * 1024 + threadIdx.x] = sum;
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}
void run_alt(int valu_steps, int atomics, int stagger) {
    const int blocks = 256, iters = 400;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, blocks * 1024 * 4); (void)hipMalloc(&cyc, blocks * 16 * 8);
    double a = 0;
    for (int rep = 0; rep < 2; ++rep) {
        alternate<<<blocks, 1024>>>(out, cyc, iters, valu_steps, atomics, stagger, 99u + rep);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 16);
        (void)hipMemcpy(h.data(), cyc, blocks * 16 * 8, hipMemcpyDeviceToHost);
        a = 0;
        for (auto v : h) a += (double)v;
        a /= h.size();
    }
    // per iteration and wave: VALU issue = valu_steps * 16 * 4 cycles (x 4 waves per SIMD); LDS = atomics * 120 cycles (x 16 waves per CU)
    printf("alternating: %4d v_fma + %2d atomics per iteration, stagger %2d: %.0f cycles per iteration;  VALU alone %d, LDS alone %d, sum %d\n", valu_steps * 16, atomics, stagger,
           a / iters, valu_steps * 16 * 4 * 4, atomics * 120 * 16, valu_steps * 16 * 4 * 4 + atomics * 120 * 16);
    (void)hipFree(out); (void)hipFree(cyc);
}

template <int OTHER>
void run_mix(const char* what, int do_atomics, int do_other) {
    const int blocks = 256, iters = 1000;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, blocks * 1024 * 4); (void)hipMalloc(&cyc, blocks * 16 * 8);
    double a = 0, o = 0;
    for (int rep = 0; rep < 2; ++rep) {
        mix<OTHER><<<blocks, 1024>>>(out, cyc, iters, do_atomics, do_other, 777u + rep);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 16);
        (void)hipMemcpy(h.data(), cyc, blocks * 16 * 8, hipMemcpyDeviceToHost);
        a = o = 0;
        for (size_t i = 0; i < h.size(); ++i) (((i >> 2) & 1) ? o : a) += (double)h[i];
        a /= h.size() / 2; o /= h.size() / 2;
    }
    printf("%-70s atomic waves %.0f cycles, other waves %.0f cycles\n", what, a, o);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    for (int st : {0, 8}) { run_alt(0, 10, st); run_alt(30, 0, st); run_alt(30, 10, st); run_alt(60, 10, st); run_alt(120, 10, st); run_alt(60, 40, st); }
    for (uint32_t a : {8u, 64u}) run<5>("pairs of atomics, random values (2 instructions per step)", a);
    run_mix<0>("8 waves: 8000 full-wave atomics each | 8 waves idle", 1, 0);
    run_mix<0>("8 waves idle | 8 waves: 240000 v_fma each", 0, 1);
    run_mix<0>("8 waves atomics | 8 waves v_fma", 1, 1);
    run_mix<1>("8 waves idle | 8 waves: 8000 sparse ds_write each", 0, 1);
    run_mix<1>("8 waves atomics | 8 waves sparse ds_write", 1, 1);
    run_mix<2>("8 waves idle | 8 waves: 8000 random global loads each", 0, 1);
    run_mix<2>("8 waves atomics | 8 waves random global loads", 1, 1);

    for (uint32_t a : {1u, 4u, 8u, 16u, 32u, 48u, 64u}) run<0>("random rows", a);
    for (uint32_t a : {8u, 32u, 64u}) run<1>("conflict-free (lane = bank)", a);
    for (uint32_t a : {8u, 32u, 64u}) run<2>("one bank, different rows", a);
    for (uint32_t a : {8u, 64u}) run<4>("plain ds_write (random rows)", a);
    for (uint32_t a : {8u, 64u}) run<5>("pairs of atomics, random values (2 per step)", a);
    return 0;
}
