// Microbenchmark of ONE environment-MLP pass exactly as the fused kernel runs it (mlp_mfma.hip.h code), in isolation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -I../../envidr_amd/csrc -I../../include -o env_pass_probe env_pass_probe.hip
#include "mlp_mfma.hip.h"
#include <cstdio>
#ifndef PROBE_T
#define PROBE_T 8
#endif
#include <vector>
using namespace envidr;

template <int PF> constexpr int ring_padded(int frags) { return (frags + PF - 1) / PF * PF; }

template <int TERMS, int ENV_T, int PF>
__global__ void __launch_bounds__(64, 1) probe(const float* __restrict__ blob, float* out, unsigned long long* cyc, int iters) {
    constexpr int kEnv0 = 0, kEnv1 = kEnv0 + lane_layer_frags(TERMS, ENV_T, true), kEnv2 = kEnv1 + tile_layer_frags(ENV_T, ENV_T, true),
                  kEnv3 = kEnv2 + tile_layer_frags(ENV_T, ENV_T, true), kEnvFrags = kEnv3 + tile_layer_frags(ENV_T, 1, true);
    constexpr uint32_t kEnvChunks = pass_chunks(kEnvFrags);
    constexpr int kEnvN = ring_padded<PF>(kEnvFrags);
    const uint32_t lane = lane_id();
    WeightRing<PF> wp;
    wp.start(nullptr, lane, 0, blob, kEnvChunks);
    float code[2 * TERMS];
#pragma unroll
    for (int s = 0; s < 2 * TERMS; ++s) code[s] = (float)(lane + s) * 1e-3f;
    float acc_out = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        f32x16 outA, outB;
#pragma unroll 1
        for (int grp = 0; grp < 2; ++grp) {
            float in[TERMS];
#pragma unroll
            for (int s = 0; s < TERMS; ++s) in[s] = grp ? code[2 * s + 1] : code[2 * s];
            f32x16 ha[ENV_T], hb[ENV_T], o[1];
            wp.begin_pass(blob, kEnvChunks, blob, kEnvChunks);
#ifndef PROBE_RELU
#define PROBE_RELU true
#endif
            pipe_layer_from_lanes<TERMS, ENV_T, kEnv0, kEnvN>(wp, lane, in, ha);
            pipe_layer_from_tiles<ENV_T, ENV_T, kEnv1, kEnvN, PROBE_RELU>(wp, lane, ha, hb);
            pipe_layer_from_tiles<ENV_T, ENV_T, kEnv2, kEnvN, PROBE_RELU>(wp, lane, hb, ha);
            pipe_layer_from_tiles<ENV_T, 1, kEnv3, kEnvN, PROBE_RELU>(wp, lane, ha, o);
            wp.template end_pass<kEnvFrags>();
            if (grp == 0) outA = o[0]; else outB = o[0];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) { acc_out += outA[r] + outB[r]; }
#pragma unroll
        for (int s = 0; s < 2 * TERMS; ++s) code[s] += acc_out * 1e-9f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = acc_out;
    if (lane == 0) { cyc[blockIdx.x] = t1 - t0; cyc[gridDim.x] = kEnvFrags; }
}

template <int PF>
void run(int blocks) {
    const int iters = 50;
    float *blob, *out; unsigned long long* cyc;
    hipMalloc(&blob, 4 << 20); hipMemset(blob, 0, 4 << 20);
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, (blocks + 1) * 8);
    for (int rep = 0; rep < 2; ++rep) {
        probe<36, PROBE_T, PF><<<blocks, 64>>>(blob, out, cyc, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks + 1);
        hipMemcpy(h.data(), cyc, (blocks + 1) * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
        const double mf = (double)h[blocks];
        if (rep) printf("env pass ring %2d, %4d waves: %.0f MFMAs per pass, %.0f ticks per pass, %.2f ticks / MFMA\n", PF, blocks, mf, avg / (iters * 2), avg / (iters * 2) / mf);
    }
    hipFree(blob); hipFree(out); hipFree(cyc);
}
int main() {
    run<32>(256);
    return 0;
}
