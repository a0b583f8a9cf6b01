// ONE environment-MLP pass (72 -> 256 -> 256 -> 256 -> 12 for a 32-sample group) exactly as the shading kernels run it
// (mlp_mfma.hip.h code), in isolation, in the variants the design chose between:
//   CLUMPS  0: the 16 B operands of an input tile (accumulator read + ReLU) as one cluster ahead of the tile's MFMAs
//           n: staged for tile K+1 in n clumps between tile K's steps (software-pipelined, two register sets)
//          -1: ReLU through the LDS atomic unit (ds_max_f32 on a zero slot), staged one tile ahead
//   OUT16   the 256 -> 12 layer on v_mfma_f32_16x16x1_4B_f32 (pipe_layer16_from_tiles) instead of a 32-row tile
//   RELU    off = the floor without any operand staging work
// plus a numerical self-test of the 16-wide layer against the host (layout of the 4-block MFMA, v_permlane16_swap).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -I../../envidr_amd/csrc -I../../include -o env_pass_probe env_pass_probe.hip
#include "env_pass.hip.h"
#include <cmath>
#include <cstdio>
#include <vector>
using namespace envidr;

namespace envidr { void set_error(const char*, ...) {} }

constexpr int kPF = 32;
constexpr int ring_padded(int frags) { return (frags + kPF - 1) / kPF * kPF; }

// the pass as the kernels run it now (env_pass.hip.h): biases from LDS, ReLU in the LDS atomic unit, 16-row output blocks
template <int TERMS, int ENV_T, bool HANDOFF>
__global__ void __launch_bounds__(64, 1) probe_env_pass(const float* __restrict__ blob, float* out, unsigned long long* cyc, int iters) {
    using L = EnvLayout<TERMS, ENV_T>;
    constexpr uint32_t kEnvChunks = pass_chunks(L::Frags);
    constexpr int kEnvN = ring_padded(L::Frags);
    const uint32_t lane = lane_id();
    __shared__ __attribute__((aligned(16))) float s_env[L::kLdsFloats];
    const EnvAux aux = env_lds_init<TERMS, ENV_T>(blob, s_env, lane);
    WeightRing<kPF> wp;
    wp.start(lane, blob, kEnvChunks);
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
            f32x16 o;
            wp.begin_pass(blob, kEnvChunks, blob, kEnvChunks);
            env_pass<TERMS, ENV_T, kEnvN, HANDOFF>(wp, lane, aux, in, o);
            if (grp == 0) outA = o; else outB = o;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_out += outA[r] + outB[r]; }
#pragma unroll
        for (int s = 0; s < 2 * TERMS; ++s) code[s] += acc_out * 1e-9f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = acc_out;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int TERMS, int ENV_T, int CLUMPS, bool OUT16, bool RELU>
__global__ void __launch_bounds__(64, 1) probe(const float* __restrict__ blob, float* out, unsigned long long* cyc, int iters) {
    constexpr int kEnv0 = 0, kEnv1 = kEnv0 + lane_layer_frags(TERMS, ENV_T, true), kEnv2 = kEnv1 + tile_layer_frags(ENV_T, ENV_T, true),
                  kEnv3 = kEnv2 + tile_layer_frags(ENV_T, ENV_T, true), kEnvFrags = kEnv3 + tile_layer_frags(ENV_T, 1, true);
    constexpr uint32_t kEnvChunks = pass_chunks(kEnvFrags);
    constexpr int kEnvN = ring_padded(kEnvFrags);
    const uint32_t lane = lane_id();
    __shared__ float lds_stage[kLdsStageFloats];
#pragma unroll
    for (int r = 0; r < 16; ++r) lds_stage[r * 64 + lane] = 0.0f;
    float* slot = lds_stage + lane;
    WeightRing<kPF> wp;
    wp.start(lane, blob, kEnvChunks);
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
            pipe_layer_from_lanes<TERMS, ENV_T, kEnv0, kEnvN>(wp, lane, in, ha);
            pipe_layer_from_tiles<ENV_T, ENV_T, kEnv1, kEnvN, RELU, true, CLUMPS>(wp, lane, ha, hb, slot);
            pipe_layer_from_tiles<ENV_T, ENV_T, kEnv2, kEnvN, RELU, true, CLUMPS>(wp, lane, hb, ha, slot);
            if constexpr (OUT16) pipe_layer16_from_tiles<ENV_T, kEnv3, kEnvN, RELU, true, (CLUMPS < 0 ? 0 : CLUMPS)>(wp, lane, ha, o[0]);
            else pipe_layer_from_tiles<ENV_T, 1, kEnv3, kEnvN, RELU, true, CLUMPS>(wp, lane, ha, o, slot);
            wp.template end_pass<kEnvFrags>();
            if (grp == 0) outA = o[0]; else outB = o[0];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_out += outA[r] + outB[r]; }
#pragma unroll
        for (int s = 0; s < 2 * TERMS; ++s) code[s] += acc_out * 1e-9f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = acc_out;
    if (lane == 0) { cyc[blockIdx.x] = t1 - t0; cyc[gridDim.x] = kEnvFrags; }
}

template <bool HANDOFF>
void run_env_pass(int blocks) {
    const int iters = 50;
    float *blob, *out; unsigned long long* cyc;
    (void)hipMalloc(&blob, 4 << 20); (void)hipMemset(blob, 0, 4 << 20);
    (void)hipMalloc(&out, blocks * 64 * 4); (void)hipMalloc(&cyc, (blocks + 1) * 8);
    for (int rep = 0; rep < 2; ++rep) {
        probe_env_pass<36, 8, HANDOFF><<<blocks, 64>>>(blob, out, cyc, iters);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks + 1);
        (void)hipMemcpy(h.data(), cyc, (blocks + 1) * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
        if (rep) printf("env_pass() of env_pass.hip.h, hand-over %d, %4d waves: %.0f ticks per pass (MFMA issue time 64 x 2368 + 32 x 129 = 155680)\n", (int)HANDOFF, blocks, avg / (iters * 2));
    }
    (void)hipFree(blob); (void)hipFree(out); (void)hipFree(cyc);
}

template <int CLUMPS, bool OUT16, bool RELU>
void run(int blocks) {
    const int iters = 50;
    float *blob, *out; unsigned long long* cyc;
    (void)hipMalloc(&blob, 4 << 20); (void)hipMemset(blob, 0, 4 << 20);
    (void)hipMalloc(&out, blocks * 64 * 4); (void)hipMalloc(&cyc, (blocks + 1) * 8);
    for (int rep = 0; rep < 2; ++rep) {
        probe<36, 8, CLUMPS, OUT16, RELU><<<blocks, 64>>>(blob, out, cyc, iters);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks + 1);
        (void)hipMemcpy(h.data(), cyc, (blocks + 1) * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
        if (rep) printf("env pass, clumps %2d, out16 %d, relu %d, %4d waves: %.0f ticks per pass (64 x 2489 = 159296; 64 x 2425 = 155200)\n",
                        CLUMPS, (int)OUT16, (int)RELU, blocks, avg / (iters * 2));
    }
    (void)hipFree(blob); (void)hipFree(out); (void)hipFree(cyc);
}

// ---- self-test of the 16-wide layer: 64 -> 12 on two groups of 32 samples, inputs given as 32x32-layout tiles --------------
__global__ void __launch_bounds__(64, 1) selftest(const float* __restrict__ blob, const float* __restrict__ x, float* __restrict__ y) {
    // x: [64 samples][64 features]; lane l of group g (= sample 32 g + (l & 31)) holds, in tile T register r, feature 32 T + tile_row(r, l >> 5)
    const uint32_t lane = lane_id();
    WeightRing<kPF> wp;
    constexpr int kFrags = tile_layer_frags(2, 1, true), kN = ring_padded(kFrags);
    wp.start(lane, blob, pass_chunks(kFrags));
    float lo[2][4], hi[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        f32x16 in[2], d;
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) in[T][r] = x[(32 * g + (lane & 31)) * 64 + 32 * T + tile_row(r, lane >> 5)];
        wp.begin_pass(blob, pass_chunks(kFrags), blob, pass_chunks(kFrags));
        pipe_layer16_from_tiles<2, 0, kN, true, true, 4>(wp, lane, in, d);
        wp.template end_pass<kFrags>();
        fold16(d, lo[g], hi[g]);
    }
    float e[12];
    rows_to_lanes<3>(lo[0], hi[0], lo[1], hi[1], e);
#pragma unroll
    for (int i = 0; i < 12; ++i) y[lane * 12 + i] = e[i];
}

int run_selftest() {
    std::vector<float> W(12 * 64), b(12), x(64 * 64), packed(kChunkFloats * 2, 0.0f), y(64 * 12);
    for (int i = 0; i < 12; ++i) { b[i] = 0.1f * (i + 1); for (int k = 0; k < 64; ++k) W[i * 64 + k] = sinf(0.37f * i + 0.11f * k) ; }
    for (int s = 0; s < 64; ++s) for (int k = 0; k < 64; ++k) x[s * 64 + k] = cosf(0.05f * s * (k + 1)) - 0.2f;
    pack_linear16(W.data(), 12, 64, packed.data(), b.data());
    float *dblob, *dx, *dy;
    (void)hipMalloc(&dblob, packed.size() * 4); (void)hipMalloc(&dx, x.size() * 4); (void)hipMalloc(&dy, y.size() * 4);
    (void)hipMemcpy(dblob, packed.data(), packed.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    selftest<<<1, 64>>>(dblob, dx, dy);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int s = 0; s < 64; ++s)
        for (int i = 0; i < 12; ++i) {
            double ref = b[i];
            for (int k = 0; k < 64; ++k) ref += (double)W[i * 64 + k] * std::fmax(x[s * 64 + k], 0.0f);
            worst = std::fmax(worst, std::fabs(ref - y[s * 12 + i]));
        }
    printf("16-wide layer self-test (64 -> 12, 64 samples, ReLU on the inputs): max |error| %.3e %s\n", worst, worst < 1e-4 ? "OK" : "FAILED");
    return worst < 1e-4 ? 0 : 1;
}

int main() {
    const int rc = run_selftest();
    run_env_pass<false>(1024);
    run_env_pass<true>(1024);
    run_env_pass<false>(1024);
    run_env_pass<true>(1024);
    run<0, false, true>(1024);      // the kernel of rounds 1-3
    run<0, false, false>(1024);     // floor without operand staging
    run<2, false, true>(1024);
    run<0, true, true>(1024);
    run<-1, false, true>(1024);     // ReLU in the LDS atomic unit
    run<-1, true, true>(1024);
    run<0, true, false>(1024);
    return rc;
}
