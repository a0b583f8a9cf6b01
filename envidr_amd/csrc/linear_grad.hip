// Weight gradient of a dense layer over a large batch: dW[o][i] = sum_m gy[m][o] x[m][i], db[o] = sum_m gy[m][o]
// (what torch autograd asks of nn.Linear in the training branch, reference nerf/render_func/cuda_ray.py:64-237 -> network.py:524-698;
// the reference leaves it to cuBLAS).  The shape is the worst case for a library GEMM -- a 256 x 256 (or 12 x 256, 256 x 72 ...) result
// reduced over 10^5 .. 10^6 samples: tiles of the RESULT are all the parallelism a conventional kernel has (32 workgroups on 256 CUs;
// the rocBLAS kernels torch picks took 0.4 ms per environment-MLP layer of a 144 k-sample training batch, 6.9 of a step's 16.2 ms).
// Here the reduction is what is split: every wave owns a 128 x 128 (or narrower) block of dW for ONE chunk of the samples and keeps it in
// its accumulator registers (16 tiles of v_mfma_f32_32x32x2_f32: A = gy^T, B = x, two samples per instruction), 1 024+ waves across the
// chip; the per-chunk partial results go to a workspace and a second kernel adds them up in a fixed order (deterministic, no atomics).
//
// Operand loads.  A lane supplies A[row = lane % 32][k = lane / 32] and B[k = lane / 32][col = lane % 32]: lane half h reads sample
// m0 + h.  WIDE operands (row length a multiple of 4, at least 64): ONE 16-byte load per lane and step, x[m][blk * 128 + 4 c .. 4 c + 3] --
// register j of that load is the operand of tile j, whose column c then stands for feature 4 c + j: four tiles per load, 512 contiguous
// bytes per half wave, and the permutation costs nothing (it is undone by the index arithmetic of the store).  NARROW operands (12, 24,
// 28 ... features): one 4-byte load, one tile, feature = blk * 32 + c.
#include "common.hip.h"
#include "../../include/envidr_render.h"

namespace envidr {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int tile_row_lg(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <bool WIDE>
struct Operand {
    static constexpr int T = WIDE ? 4 : 1;          // tiles per block
    static constexpr int W = WIDE ? 128 : 32;       // features per block
    float v[T];
    // `p` addresses this lane's feature(s) of sample 0 (clamped into the row for lanes past the layer's width: what they load only reaches
    // rows / columns of the tiles that are never stored); no bounds test, no branch: the wait for a load sits where its value is used
    __device__ __forceinline__ void load(const float* __restrict__ p, size_t row, uint32_t stride) {
        if constexpr (WIDE) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(p + row * stride);
            v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
        } else {
            v[0] = p[row * stride];
        }
    }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int t = 0; t < T; ++t) v[t] = 0.0f;
    }
    static __device__ __forceinline__ uint32_t lane_feature(uint32_t blk, uint32_t c) { return WIDE ? blk * 128 + 4 * c : blk * 32 + c; }
    // feature index of (tile t, position q inside the tile)
    static __device__ __forceinline__ uint32_t feature(uint32_t blk, int t, uint32_t q) { return WIDE ? blk * 128 + 4 * q + t : blk * 32 + q; }
};

template <int AT, int BT>
__device__ __forceinline__ void mfma_step(f32x16 (&acc)[AT][BT], float (&bsum)[AT], const float (&a)[AT], const float (&b)[BT]) {
#pragma unroll
    for (int i = 0; i < AT; ++i) {
#pragma unroll
        for (int j = 0; j < BT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        bsum[i] += a[i];
    }
}

constexpr uint32_t kGradWaves = 4;          // waves per workgroup: their chunks' results are added in LDS before anything is written

template <bool A_WIDE, bool B_WIDE>
__global__ void __launch_bounds__(64 * kGradWaves) k_linear_weight_grad(const float* __restrict__ x, const float* __restrict__ gy, uint32_t M, uint32_t K_in,
                                                                       uint32_t N_out, uint32_t rows_per_chunk, float* __restrict__ part_w,
                                                                       float* __restrict__ part_b) {
    using OpA = Operand<A_WIDE>;
    using OpB = Operand<B_WIDE>;
    constexpr int AT = OpA::T, BT = OpB::T;
    constexpr int kUnits = AT * BT * 4 + 1;                    // 16-byte units of one wave's accumulators (+ the bias sums)
    __shared__ __attribute__((aligned(16))) float s_acc[2 * kUnits * 64 * 4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, half = lane >> 5, c = lane & 31;
    const uint32_t ob = blockIdx.x, ib = blockIdx.y, group = blockIdx.z, chunk = group * kGradWaves + wave;
    const uint32_t m_begin = min(M, chunk * rows_per_chunk), m_end = min(M, m_begin + rows_per_chunk);      // (a wave past the last chunk: no rows, zeros)
    f32x16 acc[AT][BT];
#pragma unroll
    for (int a = 0; a < AT; ++a)
#pragma unroll
        for (int b = 0; b < BT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;
    float bsum[AT];
#pragma unroll
    for (int a = 0; a < AT; ++a) bsum[a] = 0;

    const uint32_t fa = OpA::lane_feature(ob, c), fb = OpB::lane_feature(ib, c);
    const float* pa = gy + (fa < N_out ? fa : 0u);
    const float* pb = x + (fb < K_in ? fb : 0u);
    // steps of two samples (lane half h: sample m0 + h); `full` of them have both samples inside the chunk.  Two register sets alternate:
    // the loads of step s + 1 are in flight under the 16 MFMAs of step s
    const uint32_t full = (m_end - m_begin) / 2;
    OpA a0, a1;
    OpB b0, b1;
    size_t m = (size_t)m_begin + half;
    if (full > 0) { a0.load(pa, m, N_out); b0.load(pb, m, K_in); }
    uint32_t sdone = 0;
    const size_t m_last = (size_t)m_end - 1;
    for (; sdone + 2 <= full; sdone += 2) {
        a1.load(pa, m + 2, N_out); b1.load(pb, m + 2, K_in);
        mfma_step<AT, BT>(acc, bsum, a0.v, b0.v);
        // (unconditional, row clamped into the chunk: a load under a branch makes the compiler wait for EVERY load in flight at the join)
        const size_t mp = m + 4 < m_last ? m + 4 : m_last;
        a0.load(pa, mp, N_out); b0.load(pb, mp, K_in);
        mfma_step<AT, BT>(acc, bsum, a1.v, b1.v);
        m += 4;
    }
    if (sdone < full) { mfma_step<AT, BT>(acc, bsum, a0.v, b0.v); m += 2; }
    if ((m_end - m_begin) & 1u) {                                // an odd chunk: the last sample belongs to lane half 0, half 1 contributes zeros
        if (half == 0) { a0.load(pa, m, N_out); b0.load(pb, m, K_in); } else { a0.zero(); b0.zero(); }
        mfma_step<AT, BT>(acc, bsum, a0.v, b0.v);
    }
    // The four waves' results meet in LDS, added in a fixed order -- (w0 + w2) + (w1 + w3) -- so one partial result per WORKGROUP goes to
    // the workspace: a quarter of the bytes the reduction kernel reads back (64 MiB for a 256 x 256 layer before: 11 us of pure HBM time
    // per layer, 41 layers a step).  Two slots of [16-byte unit][lane]: 16-byte LDS accesses, consecutive lanes consecutive units.
    auto put = [&](uint32_t slot) {
        f32x4* dst = reinterpret_cast<f32x4*>(s_acc) + (size_t)slot * kUnits * 64 + lane;
#pragma unroll
        for (int a = 0; a < AT; ++a)
#pragma unroll
            for (int b = 0; b < BT; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[((a * BT + b) * 4 + q) * 64] = f32x4{acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
        f32x4 bs = {0, 0, 0, 0};
#pragma unroll
        for (int a = 0; a < AT; ++a) bs[a] = bsum[a];
        dst[AT * BT * 4 * 64] = bs;
    };
    auto add = [&](uint32_t slot) {
        const f32x4* src = reinterpret_cast<const f32x4*>(s_acc) + (size_t)slot * kUnits * 64 + lane;
#pragma unroll
        for (int a = 0; a < AT; ++a)
#pragma unroll
            for (int b = 0; b < BT; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = src[((a * BT + b) * 4 + q) * 64];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[a][b][4 * q + i] += v[i];
                }
        const f32x4 bs = src[AT * BT * 4 * 64];
#pragma unroll
        for (int a = 0; a < AT; ++a) bsum[a] += bs[a];
    };
    if (wave >= 2) put(wave - 2);
    __syncthreads();
    if (wave < 2) add(wave);
    __syncthreads();
    if (wave == 1) put(0);
    __syncthreads();
    if (wave == 0) add(0);
    if (wave != 0) return;
    // partial dW of this group: acc[a][b][r] of lane (half, c) = dW[o = featA(a, tile_row(r, half))][i = featB(b, c)]
    float* pw = part_w + (size_t)group * N_out * K_in;
#pragma unroll
    for (int a = 0; a < AT; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t o = OpA::feature(ob, a, (uint32_t)tile_row_lg(r, (int)half));
            if (o >= N_out) continue;
            if constexpr (B_WIDE) {
                const uint32_t i = ib * 128 + 4 * c;              // the four tiles' values of this lane are four consecutive inputs: one 16-byte store
                if (i < K_in) *reinterpret_cast<f32x4*>(pw + (size_t)o * K_in + i) = f32x4{acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
            } else {
                const uint32_t i = ib * 32 + c;
                if (i < K_in) pw[(size_t)o * K_in + i] = acc[a][0][r];
            }
        }
    // partial db: a lane summed gy[.][featA(a, c)] over its half's samples; the two halves hold the same features
    if (part_b && ib == 0) {
#pragma unroll
        for (int a = 0; a < AT; ++a) {
            const float other = __shfl_xor(bsum[a], 32);
            const uint32_t o = OpA::feature(ob, a, c);
            if (half == 0 && o < N_out) part_b[(size_t)group * N_out + o] = bsum[a] + other;
        }
    }
}

// dst[e] (+)= sum over chunks of part[chunk][e].  64 consecutive elements per workgroup, the chunks dealt round-robin to its four waves,
// eight loads in flight per lane; the four waves' sums meet in LDS and are added in wave order: a fixed order, whatever the launch does
__global__ void __launch_bounds__(kBlock) k_reduce_partials(const float* __restrict__ part, uint32_t chunks, uint32_t elems, int accumulate,
                                                            float* __restrict__ dst) {
    __shared__ float s_sum[4][64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t e = blockIdx.x * 64 + lane;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (e < elems) {
        uint32_t k = wave;
        for (; k + 28 < chunks; k += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += part[(size_t)(k + 4 * u) * elems + e];
        }
        for (int u = 0; k < chunks; k += 4, ++u) acc[u & 7] += part[(size_t)k * elems + e];
    }
    s_sum[wave][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (wave == 0 && e < elems) {
        const float s = (s_sum[0][lane] + s_sum[1][lane]) + (s_sum[2][lane] + s_sum[3][lane]);
        dst[e] = accumulate ? dst[e] + s : s;
    }
}

struct GradPlan { uint32_t blocks_o, blocks_i, chunks, groups, rows_per_chunk; bool a_wide, b_wide; };

GradPlan plan_weight_grad(uint32_t M, uint32_t K_in, uint32_t N_out) {
    GradPlan p;
    p.a_wide = N_out >= 64 && N_out % 4 == 0;
    p.b_wide = K_in >= 64 && K_in % 4 == 0;
    p.blocks_o = ceil_div(N_out, p.a_wide ? 128u : 32u);
    p.blocks_i = ceil_div(K_in, p.b_wide ? 128u : 32u);
    // one wave per SIMD (the wide-wide form holds 256 accumulator registers): 4 waves per CU, all resident at once
    const uint32_t waves = 4u * 256u;
    uint32_t chunks = std::max(1u, waves / (p.blocks_o * p.blocks_i));
    chunks = std::min(chunks, std::max(1u, M / 128u));         // at least 128 samples per chunk: the partial store must stay small beside the MFMAs
    p.rows_per_chunk = (ceil_div(M, chunks) + 1u) & ~1u;       // even: a step is two samples
    p.chunks = ceil_div(M, p.rows_per_chunk);
    p.groups = ceil_div(p.chunks, kGradWaves);                 // one partial result per workgroup of four waves
    return p;
}

}  // namespace
}  // namespace envidr

using namespace envidr;

extern "C" {

uint64_t envidr_linear_weight_grad_workspace_bytes(uint32_t M, uint32_t K_in, uint32_t N_out) {
    if (M == 0 || K_in == 0 || N_out == 0) return 0;
    const GradPlan p = plan_weight_grad(M, K_in, N_out);
    return (uint64_t)p.groups * ((uint64_t)N_out * K_in + N_out) * sizeof(float);
}

int envidr_linear_weight_grad(const float* x, const float* gy, uint32_t M, uint32_t K_in, uint32_t N_out, float* dW, float* db, int accumulate,
                              void* workspace, uint64_t workspace_bytes, envidr_stream_t stream) {
    ENVIDR_REQUIRE(K_in >= 1 && N_out >= 1 && dW, "linear_weight_grad: empty layer / null dW");
    hipStream_t s = as_stream(stream);
    if (M == 0) {
        if (!accumulate) {
            if (hipMemsetAsync(dW, 0, (size_t)N_out * K_in * 4, s) != hipSuccess) return check_launch("linear_weight_grad memset");
            if (db && hipMemsetAsync(db, 0, (size_t)N_out * 4, s) != hipSuccess) return check_launch("linear_weight_grad memset");
        }
        return ENVIDR_OK;
    }
    ENVIDR_REQUIRE(x && gy, "linear_weight_grad: null pointer");
    ENVIDR_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(gy) & 15) == 0, "linear_weight_grad: x / gy must be 16-byte aligned");
    const GradPlan p = plan_weight_grad(M, K_in, N_out);
    const uint64_t need = envidr_linear_weight_grad_workspace_bytes(M, K_in, N_out);
    ENVIDR_REQUIRE(workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
                   "linear_weight_grad: workspace of %llu bytes (16-byte aligned) needed, %llu given", (unsigned long long)need,
                   (unsigned long long)workspace_bytes);
    float* part_w = static_cast<float*>(workspace);
    float* part_b = part_w + (size_t)p.groups * N_out * K_in;
    const dim3 grid(p.blocks_o, p.blocks_i, p.groups), block(64 * kGradWaves);
#define ENVIDR_WG(AW, BW) hipLaunchKernelGGL((k_linear_weight_grad<AW, BW>), grid, block, 0, s, x, gy, M, K_in, N_out, p.rows_per_chunk, part_w, db ? part_b : nullptr)
    if (p.a_wide && p.b_wide) ENVIDR_WG(true, true);
    else if (p.a_wide) ENVIDR_WG(true, false);
    else if (p.b_wide) ENVIDR_WG(false, true);
    else ENVIDR_WG(false, false);
#undef ENVIDR_WG
    int rc = check_launch("k_linear_weight_grad");
    if (rc) return rc;
    hipLaunchKernelGGL(k_reduce_partials, dim3(ceil_div(N_out * K_in, 64u)), dim3(kBlock), 0, s, part_w, p.groups, N_out * K_in, accumulate, dW);
    rc = check_launch("k_reduce_partials");
    if (rc || !db) return rc;
    hipLaunchKernelGGL(k_reduce_partials, dim3(ceil_div(N_out, 64u)), dim3(kBlock), 0, s, part_b, p.groups, N_out, accumulate, db);
    return check_launch("k_reduce_partials");
}

}  // extern "C"
