// Random-gather ceilings on gfx950: 8-byte (and 16-byte) gathers from a table of a given size, one row per lane per
// load, lanes independent (every lane its own cache line) or grouped (G consecutive lanes share a 64-byte line).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -o gather_probe gather_probe.hip && ./gather_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// rows: table rows of 8 bytes (power of two); group: lanes per shared line (1, 4, 16, 64); WIDE: 16-byte loads
template <bool WIDE>
__global__ void __launch_bounds__(256) k_gather(const float* __restrict__ table, uint32_t rows_mask, uint32_t group, uint32_t iters, float* out) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(table), 0, (int)((rows_mask + 1) * 8u), 0x00020000);
    float acc = 0;
    uint32_t seed = mix(gid / group + 12345u);
    const uint32_t sub = gid % group;
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            seed = mix(seed + 0x9e3779b9u * (k + 1));
            // the group's lanes land in one 64-byte line (8 rows): line chosen by the group's seed, row inside by the lane
            r[k] = group > 1 ? ((seed & rows_mask & ~7u) | (sub & 7u)) : (seed & rows_mask);
        }
        if constexpr (WIDE) {
            u32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (r[k] & ~1u) * 8u, 0, 0);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += __uint_as_float(v[k][0]) + __uint_as_float(v[k][3]);
        } else {
            u32x2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b64(rs, r[k] * 8u, 0, 0);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += __uint_as_float(v[k][0]) + __uint_as_float(v[k][1]);
        }
    }
    if (acc == 123.456f) out[gid] = acc;
}

int main() {
    const size_t max_bytes = 1ull << 30;
    float* table; float* out;
    hipMalloc(&table, max_bytes); hipMalloc(&out, 1 << 24);
    hipMemset(table, 0, max_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%8s %6s %5s %6s | %10s %12s %12s\n", "table", "waves", "group", "width", "ms", "Glookups/s", "lines TB/s");
    for (size_t mb : {1, 4, 16, 48, 256, 1024}) {
        size_t rows = 1; while (rows * 2 * 8 <= mb * (1ull << 20)) rows *= 2;
        for (uint32_t waves_per_cu : {8u, 16u, 32u})
            for (uint32_t group : {1u, 4u, 64u})
                for (int wide = 0; wide < 2; ++wide) {
                    if (wide && group != 1) continue;
                    const uint32_t blocks = 256 * waves_per_cu / 4, iters = 64;
                    auto launch = [&]() {
                        if (wide) hipLaunchKernelGGL(k_gather<true>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)rows - 1, group, iters, out);
                        else hipLaunchKernelGGL(k_gather<false>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)rows - 1, group, iters, out);
                    };
                    launch(); hipDeviceSynchronize();
                    hipEventRecord(e0); for (int r = 0; r < 3; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
                    const double lookups = (double)blocks * 256 * iters * 8;
                    const double lines = lookups / group;
                    printf("%6zuMB %6u %5u %6s | %10.3f %12.1f %12.2f\n", rows * 8 >> 20, waves_per_cu, group, wide ? "16B" : "8B", ms, lookups / ms / 1e6, lines * 64 / ms / 1e9);
                }
    }
    return 0;
}
