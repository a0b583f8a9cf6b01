// What does ONE vector-memory instruction cost a wave that is otherwise issuing v_mfma_f32_32x32x16_f16 back to back
// (one wave per SIMD, 4 waves per workgroup)?  One instruction of KIND every EVERY MFMAs, all hitting L2 (a 64-KiB window).
//   KIND 0 none | 1 buffer_load_dwordx4 -> VGPR | 2 buffer_load_dword -> VGPR | 3 buffer_load_dwordx4 ... lds (M0 saved/restored)
//        4 buffer_load_dwordx4 ... lds (M0 set once, outside) | 5 buffer_load_dword ... lds | 6 global_load_lds_dwordx4 | 7 ds_read_b128
//        8 ds_write_b128 | 9 = 3 but streaming a 640-KiB blob (all workgroups the same one: L2 hits, no vector-L1 hits)
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -o mfma_vmem_probe mfma_vmem_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int KIND, int EVERY>
__global__ void __launch_bounds__(256, 1) probe(const float* __restrict__ src, float* out, unsigned long long* cyc, int iters) {
    __shared__ u32x4 s_buf[4096];            // 64 KiB
    const unsigned lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    half8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(lane * 1e-3f); b[j] = (_Float16)(j * 1e-3f); }
    const uint64_t addr = (uint64_t)src;
    u32x4 rsrc = {(uint32_t)addr, (uint32_t)(addr >> 32) & 0xffffu, 1u << 20, 0x00020000u};
#pragma unroll
    for (int q = 0; q < 4; ++q) rsrc[q] = __builtin_amdgcn_readfirstlane(rsrc[q]);
    const uint32_t voff = (wave * 64u + lane) * 16u;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)s_buf) + wave * 16384u;
    u32x4 sink = {0, 0, 0, 0};
    uint32_t sink1 = 0;
    if (KIND == 4) asm volatile("s_mov_b32 m0, %0" ::"s"(lds0));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const uint32_t soff = KIND == 9 ? (uint32_t)((it * 16 + s) % 160) * 4096u : (uint32_t)(s & 15) * 4096u, ldst = lds0 + (uint32_t)(s & 15) * 1024u;
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                acc[q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q & 1], 0, 0, 0);
                if ((s * 6 + q) % EVERY == EVERY - 1) {
                    if constexpr (KIND == 1) { u32x4 v; asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory"); sink = v; }
                    if constexpr (KIND == 2) { uint32_t v; asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory"); sink1 = v; }
                    if constexpr (KIND == 3 || KIND == 9) { uint32_t keep; asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(ldst), "v"(voff), "s"(rsrc), "s"(soff) : "memory"); }
                    if constexpr (KIND == 4) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff) : "memory");
                    if constexpr (KIND == 5) { uint32_t keep; asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(ldst), "v"(voff), "s"(rsrc), "s"(soff) : "memory"); }
                    if constexpr (KIND == 6) { uint32_t keep; asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(ldst), "v"(voff), "s"(src) : "memory"); }
                    if constexpr (KIND == 7) { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(voff) : "memory"); sink = v; }
                    if constexpr (KIND == 8) { asm volatile("ds_write_b128 %0, %1" ::"v"(voff), "v"(sink) : "memory"); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (KIND == 1 || KIND == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if (KIND == 7 || KIND == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = (float)(sink[0] + sink[1] + sink[2] + sink[3] + sink1) + (float)s_buf[threadIdx.x][0];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int EVERY>
void run(const char* what) {
    const int blocks = 256, iters = 100;
    float *out, *src; unsigned long long* cyc;
    (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&cyc, blocks * 8); (void)hipMalloc(&src, 2 << 20); (void)hipMemset(src, 0, 2 << 20);
    for (int rep = 0; rep < 2; ++rep) {
        probe<KIND, EVERY><<<blocks, 256>>>(src, out, cyc, iters);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        const double per = avg / (iters * 96.0);
        if (rep) printf("%-44s every %d MFMAs: %.2f ticks / MFMA  (%+.1f ticks per instruction)\n", what, EVERY, per, (per - 32.2) * EVERY);
    }
    (void)hipFree(out); (void)hipFree(cyc); (void)hipFree(src);
}
int main() {
    run<0, 6>("nothing");
    run<1, 6>("buffer_load_dwordx4 -> VGPR"); run<1, 2>("buffer_load_dwordx4 -> VGPR");
    run<2, 6>("buffer_load_dword -> VGPR"); run<2, 2>("buffer_load_dword -> VGPR");
    run<3, 6>("buffer_load_dwordx4 lds (M0 saved/restored)"); run<3, 2>("buffer_load_dwordx4 lds (M0 saved/restored)");
    run<4, 6>("buffer_load_dwordx4 lds (M0 fixed)"); run<4, 2>("buffer_load_dwordx4 lds (M0 fixed)");
    run<5, 6>("buffer_load_dword lds"); run<5, 2>("buffer_load_dword lds");
    run<6, 6>("global_load_lds_dwordx4"); run<6, 2>("global_load_lds_dwordx4");
    run<9, 12>("buffer_load_dwordx4 lds, 640-KiB stream"); run<9, 6>("buffer_load_dwordx4 lds, 640-KiB stream"); run<9, 3>("buffer_load_dwordx4 lds, 640-KiB stream");
    run<7, 6>("ds_read_b128"); run<7, 2>("ds_read_b128"); run<7, 1>("ds_read_b128");
    run<8, 6>("ds_write_b128"); run<8, 2>("ds_write_b128");
    return 0;
}
