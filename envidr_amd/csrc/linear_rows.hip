// A dense layer over a large batch of rows: y[m][o] = epilogue(sum_i x[m][i] W(o, i) (+ b[o])), M = 10^4 .. 10^6 rows, layer widths <= a few
// hundred -- the forward and input-gradient products of the training branch (what torch autograd asks of nn.Linear, reference
// nerf/render_func/cuda_ray.py:64-237 -> network.py:524-698; the reference leaves them to cuBLAS plus one elementwise kernel per bias /
// ReLU / ReLU-gradient).  fp32 on the matrix cores (v_mfma_f32_32x32x2_f32), like envidr_linear_weight_grad (linear_grad.hip) and the
// frame's shading kernels.
//
// Work decomposition.  A workgroup of four waves owns 128 rows and a block of up to 128 output columns; every wave keeps its 32 rows x (NT x 32)
// columns in accumulator registers.  W is walked in slabs of 16 inputs: a slab ([16][NT x 32] floats, <= 16 KiB, zero-filled past the layer's
// widths) is fetched from global memory -- L2: all workgroups read the same <= 256 KiB -- into registers while the previous slab is being
// multiplied, parked in the other half of a double-buffered LDS array, one barrier per slab.  W is addressed by two strides, so W [N][K]
// (forward: y = x W^T) and its transpose (input gradient: gx = gy W, no materialised W^T) take the same kernel; the staging loads are 16
// bytes wide along whichever index is contiguous.
// Operands.  Lane (h = lane / 32, c = lane % 32) supplies A[row c][k = h] and B[k = h][column c].  Its x operand for FOUR consecutive
// steps is one 16-byte LDS read x[row][k8 + 4 h .. 4 h + 3] -- step i then multiplies inputs {k8 + i, k8 + 4 + i}, an order of summation the
// B side simply follows (LDS row 8 kc + 4 h + i); the rows reach LDS by coalesced loads (RowStage).
// Epilogues (one pass over the accumulators, nothing else touches y): + bias; + bias then ReLU (the activation is what is stored: the
// backward pass needs only its sign); x (act > 0) -- the ReLU gradient folded into the product that produces the incoming gradient.
#include "common.hip.h"
#include "../../include/envidr_render.h"

namespace envidr {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSlab = 16;                        // inputs per LDS slab
constexpr int kRowsThreads = 256;                // 4 waves x 32 rows
enum WLayout { W_IN_CONTIG = 0, W_OUT_CONTIG = 1, W_ANY = 2 };

__host__ __device__ constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int NT, int WL, bool FULL>
struct SlabStage {
    static constexpr int kCols = NT * 32;
    static constexpr int kUnits = kSlab * kCols / 4;                 // 16-byte units per slab
    static constexpr int U = (kUnits + kRowsThreads - 1) / kRowsThreads;
    f32x4 q[U];
    bool ok[U][4];          // (element-wise only for W_ANY; the vector layouts use [u][0])
    uint32_t off[U];        // FULL: this thread's units as offsets from the slab's (wave-uniform) base
    // FULL (K a multiple of the slab, N of the column block: every layer of the shipped networks but the 72- and 12-wide ends): the address of a
    // unit is a uniform base, advanced on the scalar unit, plus a per-thread offset fixed for the whole kernel -- no vector instruction per
    // slab besides the loads themselves.  (An fp32 MFMA keeps its SIMD's vector ALU busy: every v_add / v_min / v_cndmask between two of them
    // is matrix time lost; with the clamped addresses and zero-selects of the general path below that was ~18 % of a 256 x 256 layer.)
    __device__ __forceinline__ void init(int64_t s_out, int64_t s_in, uint32_t n0, uint32_t tid) {
        if constexpr (FULL) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t e = min(tid + (uint32_t)u * kRowsThreads, (uint32_t)kUnits - 1u);
                if constexpr (WL == W_IN_CONTIG) off[u] = (uint32_t)((n0 + e % kCols) * s_out + 4u * (e / kCols));
                else if constexpr (WL == W_OUT_CONTIG) off[u] = (uint32_t)((e / (kCols / 4)) * s_in + n0 + 4u * (e % (kCols / 4)));
            }
        }
    }
    // General path: every load is unconditional at an address clamped into the matrix (a load under a branch drains all loads in flight at the
    // join); what lies past the layer's widths is replaced by zeros when the slab is parked in LDS -- not here: a select right behind the load
    // would make the wave wait for it before the MFMAs the load is meant to hide under
    __device__ __forceinline__ void load(const float* __restrict__ W, int64_t s_out, int64_t s_in, uint32_t k0, uint32_t K, uint32_t n0, uint32_t N,
                                         uint32_t tid) {
        if constexpr (FULL && WL != W_ANY) {
            const float* base = W + (WL == W_IN_CONTIG ? (int64_t)min(k0, K - kSlab) : (int64_t)min(k0, K - kSlab) * s_in);      // (uniform; past the end: the last slab again, unused)
#pragma unroll
            for (int u = 0; u < U; ++u) { q[u] = *reinterpret_cast<const f32x4*>(base + off[u]); ok[u][0] = true; }
            return;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t e = min(tid + (uint32_t)u * kRowsThreads, (uint32_t)kUnits - 1u);
            if constexpr (WL == W_IN_CONTIG) {
                const uint32_t j = e % kCols, k = k0 + 4u * (e / kCols), col = n0 + j;
                const f32x4 v = *reinterpret_cast<const f32x4*>(W + (int64_t)min(col, N - 1u) * s_out + min(k, K - 4u));
                q[u] = v; ok[u][0] = col < N && k < K;
            } else if constexpr (WL == W_OUT_CONTIG) {
                const uint32_t jq = e % (kCols / 4), k = k0 + e / (kCols / 4), col = n0 + 4u * jq;
                const f32x4 v = *reinterpret_cast<const f32x4*>(W + (int64_t)min(k, K - 1u) * s_in + min(col, N - 4u));
                q[u] = v; ok[u][0] = col < N && k < K;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t idx = 4u * e + i, j = idx % kCols, k = k0 + idx / kCols, col = n0 + j;
                    const float v = W[(int64_t)min(col, N - 1u) * s_out + (int64_t)min(k, K - 1u) * s_in];
                    q[u][i] = v; ok[u][i] = col < N && k < K;
                }
            }
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ slab, uint32_t tid) const {
        constexpr bool kPlain = FULL && WL != W_ANY;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t e = tid + (uint32_t)u * kRowsThreads;
            if (kUnits % kRowsThreads != 0 && e >= (uint32_t)kUnits) continue;
            if constexpr (WL == W_IN_CONTIG) {
                const uint32_t j = e % kCols, kq = e / kCols;
#pragma unroll
                for (int i = 0; i < 4; ++i) slab[(4 * kq + i) * kCols + j] = (kPlain || ok[u][0]) ? q[u][i] : 0.0f;
            } else if constexpr (WL == W_OUT_CONTIG) {
                *reinterpret_cast<f32x4*>(slab + 4u * e) = (kPlain || ok[u][0]) ? q[u] : f32x4{0, 0, 0, 0};            // e = k * (kCols / 4) + jq
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) slab[4u * e + i] = ok[u][i] ? q[u][i] : 0.0f;
            }
        }
    }
};

// The x rows of a slab: 32 rows x 16 inputs = 32 pieces of 64 bytes per wave.  Loaded with four lanes per row piece (two 16-byte loads per lane,
// rows l / 4 and 16 + l / 4: every group of four lanes covers one 64-byte piece, 16 pieces per instruction) and handed to the lanes that need
// them as MFMA operands through the wave's own corner of LDS.  (Loading the operand layout directly -- lane c reads row c -- makes every group of
// four lanes touch four different lines: the texture addresser then works 4x longer per instruction, and with eight waves per CU it was busy
// 3/4 of the time: measured 40 us of a 220 us layer.)
constexpr int kRowPitch = kSlab + 4;             // floats: 16-byte aligned rows, conflict-free 16-byte reads down a column of rows
template <bool FULL>
struct RowStage {
    f32x4 v[2];
    bool ok;
    uint32_t off[2];        // FULL: row piece offsets from x + k0
    __device__ __forceinline__ void init(uint32_t ldx, uint32_t row0, uint32_t M, uint32_t lane) {
        if constexpr (FULL) {
#pragma unroll
            for (int p = 0; p < 2; ++p) off[p] = min(row0 + 16u * p + (lane >> 2), M - 1u) * ldx + 4u * (lane & 3u);
        }
    }
    __device__ __forceinline__ void load(const float* __restrict__ x, uint32_t ldx, uint32_t row0, uint32_t M, uint32_t k0, uint32_t K, uint32_t lane) {
        if constexpr (FULL) {
            const float* base = x + min(k0, K - kSlab);
#pragma unroll
            for (int p = 0; p < 2; ++p) v[p] = *reinterpret_cast<const f32x4*>(base + off[p]);
            ok = true;
            return;
        }
        const uint32_t k = k0 + 4u * (lane & 3u);
        ok = k < K;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const uint32_t row = min(row0 + 16u * p + (lane >> 2), M - 1u);
            v[p] = *reinterpret_cast<const f32x4*>(x + (size_t)row * ldx + min(k, K - 4u));
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ rows, uint32_t lane) const {          // rows: this wave's [32][kRowPitch]
#pragma unroll
        for (int p = 0; p < 2; ++p)
            *reinterpret_cast<f32x4*>(rows + (16u * p + (lane >> 2)) * kRowPitch + 4u * (lane & 3u)) = (FULL || ok) ? v[p] : f32x4{0, 0, 0, 0};
    }
};

struct RowsArgs {
    const float* x; const float* W; const float* bias; const float* act; float* y;
    int64_t w_s_out, w_s_in;
    uint32_t ldx, ldact, ldy, M, K, N;
};

template <int NT, int WL, int EPI, bool FULL>
__global__ void __launch_bounds__(kRowsThreads, 2) k_linear_rows(RowsArgs a) {
    constexpr int kCols = NT * 32;
    __shared__ __attribute__((aligned(16))) float s_w[2][kSlab * kCols];
    __shared__ __attribute__((aligned(16))) float s_x[2][4][32 * kRowPitch];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, h = lane >> 5, c = lane & 31u;
    const uint32_t row0 = blockIdx.x * 128u + wave * 32u, n0 = blockIdx.y * kCols;
    const uint32_t K = a.K, N = a.N, M = a.M;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;

    SlabStage<NT, WL, FULL> st;
    st.init(a.w_s_out, a.w_s_in, n0, tid);
    // x comes from HBM (~2 us away under load; a slab's MFMAs take 1.7 us): the rows of three slabs are in flight per wave -- with one, the
    // layer's time was the SUM of its MFMA time and its memory time
    RowStage<FULL> x1, x2, x3;
    x1.init(a.ldx, row0, M, lane);
    x2.init(a.ldx, row0, M, lane);
    x3.init(a.ldx, row0, M, lane);
    st.load(a.W, a.w_s_out, a.w_s_in, 0, K, n0, N, tid);
    x1.load(a.x, a.ldx, row0, M, 0, K, lane);
    st.store(s_w[0], tid);
    x1.store(s_x[0][wave], lane);
    x1.load(a.x, a.ldx, row0, M, kSlab, K, lane);
    x2.load(a.x, a.ldx, row0, M, 2 * kSlab, K, lane);
    __syncthreads();
    const uint32_t slabs = (K + kSlab - 1) / kSlab;
    // one slab: `fetch` receives the rows of slab s + 3, `park` (loaded two slabs ago) holds those of slab s + 1 and goes to LDS behind the MFMAs.
    // The three register sets take these roles in turn -- the loop is unrolled by three instead of copying one set into the next: a copy
    // of a register that a load is still filling waits for the load, which is the prefetch undone
    auto slab = [&](uint32_t s, RowStage<FULL>& fetch, const RowStage<FULL>& park) {
        // the next slab of W and the rows three slabs ahead are in flight under this slab's MFMAs (past the end: clamped addresses, zeros, unused)
        st.load(a.W, a.w_s_out, a.w_s_in, (s + 1) * kSlab, K, n0, N, tid);
        fetch.load(a.x, a.ldx, row0, M, (s + 3) * kSlab, K, lane);
        // this lane's operands of the slab's eight steps: x[row c][8 kc + 4 h + i]
        f32x4 xa[kSlab / 8];
#pragma unroll
        for (int kc = 0; kc < kSlab / 8; ++kc) xa[kc] = *reinterpret_cast<const f32x4*>(s_x[s & 1u][wave] + c * kRowPitch + 8 * kc + 4 * h);
        __builtin_amdgcn_sched_barrier(0);          // (the loads stay in front of the MFMAs, their uses behind them)
        const float* w = s_w[s & 1u] + 4u * h * kCols + c;
        // the B values of step j + 1 are read from LDS while the MFMAs of step j run (an LDS read takes about as long as two MFMAs: read,
        // wait, multiply in source order leaves the matrix core idle half the time)
        float bv[2][NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[0][t] = w[32 * t];
#pragma unroll
        for (int j = 0; j < kSlab / 2; ++j) {
            if (j + 1 < kSlab / 2) {
#pragma unroll
                for (int t = 0; t < NT; ++t) bv[(j + 1) & 1][t] = w[(8 * ((j + 1) >> 2) + ((j + 1) & 3)) * kCols + 32 * t];
            }
            const float av = xa[j >> 2][j & 3];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j & 1][t], acc[t], 0, 0, 0);
        }
        // the order above, imposed on the scheduler (left alone it sinks every read to just in front of its MFMA): the reads of step 0, then
        // per step the reads of the next one followed by this one's MFMAs (reads pair up into ds_read2_b32)
        __builtin_amdgcn_sched_group_barrier(0x100, (NT + 1) / 2, 0);
#pragma unroll
        for (int j = 0; j < kSlab / 2; ++j) {
            if (j + 1 < kSlab / 2) __builtin_amdgcn_sched_group_barrier(0x100, (NT + 1) / 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
        }
        __builtin_amdgcn_sched_barrier(0);          // nothing that waits for the loads above moves in front of the MFMAs
        st.store(s_w[(s + 1u) & 1u], tid);
        park.store(s_x[(s + 1u) & 1u][wave], lane);
        __syncthreads();
    };
    for (uint32_t s = 0;; s += 3) {
        slab(s, x3, x1);
        if (s + 1 >= slabs) break;
        slab(s + 1, x1, x2);
        if (s + 2 >= slabs) break;
        slab(s + 2, x2, x3);
        if (s + 3 >= slabs) break;
    }

    // acc[t][r] of lane (h, c) = y[row0 + acc_row(r, h)][n0 + 32 t + c]: per r the 32 lanes of a half write 128 contiguous bytes.
    // (a wave whose 32 rows all exist -- every wave but the batch's last -- stores without a branch per row)
    const bool full = row0 + 32u <= M;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const uint32_t col = n0 + 32u * t + c;
        if (col >= N) continue;
        float b = 0.0f;
        if constexpr (EPI == ENVIDR_ROWS_BIAS || EPI == ENVIDR_ROWS_BIAS_RELU) b = a.bias[col];
        float v[16];
        if constexpr (EPI == ENVIDR_ROWS_RELU_MASK) {
            float m[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) m[r] = a.act[(size_t)min(row0 + (uint32_t)acc_row(r, (int)h), M - 1u) * a.ldact + col];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = m[r] > 0.0f ? acc[t][r] : 0.0f;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v[r] = EPI == ENVIDR_ROWS_PLAIN ? acc[t][r] : acc[t][r] + b;
                if constexpr (EPI == ENVIDR_ROWS_BIAS_RELU) v[r] = v[r] < 0.0f ? 0.0f : v[r];     // (this way round a NaN stays a NaN, as in torch.relu: a diverged network shows)
            }
        }
        float* yc = a.y + col;
        if (full) {
#pragma unroll
            for (int r = 0; r < 16; ++r) yc[(size_t)(row0 + (uint32_t)acc_row(r, (int)h)) * a.ldy] = v[r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t row = row0 + (uint32_t)acc_row(r, (int)h);
                if (row < M) yc[(size_t)row * a.ldy] = v[r];
            }
        }
    }
}

template <int NT, int WL, bool FULL>
int launch_rows_epi(const RowsArgs& a, int epilogue, dim3 grid, hipStream_t s) {
    switch (epilogue) {
        case ENVIDR_ROWS_PLAIN: hipLaunchKernelGGL((k_linear_rows<NT, WL, ENVIDR_ROWS_PLAIN, FULL>), grid, dim3(kRowsThreads), 0, s, a); break;
        case ENVIDR_ROWS_BIAS: hipLaunchKernelGGL((k_linear_rows<NT, WL, ENVIDR_ROWS_BIAS, FULL>), grid, dim3(kRowsThreads), 0, s, a); break;
        case ENVIDR_ROWS_BIAS_RELU: hipLaunchKernelGGL((k_linear_rows<NT, WL, ENVIDR_ROWS_BIAS_RELU, FULL>), grid, dim3(kRowsThreads), 0, s, a); break;
        default: hipLaunchKernelGGL((k_linear_rows<NT, WL, ENVIDR_ROWS_RELU_MASK, FULL>), grid, dim3(kRowsThreads), 0, s, a); break;
    }
    return check_launch("k_linear_rows");
}

template <int NT>
int launch_rows_layout(const RowsArgs& a, int layout, int epilogue, dim3 grid, hipStream_t s) {
    // the fast addressing: whole slabs, whole column blocks, 32-bit element offsets
    const bool full = a.K % kSlab == 0 && a.N % (NT * 32u) == 0 && (uint64_t)a.M * a.ldx < (1ull << 30) &&
                      (uint64_t)a.N * (uint64_t)a.w_s_out + (uint64_t)a.K * (uint64_t)a.w_s_in < (1ull << 30);
    if (layout == W_IN_CONTIG) return full ? launch_rows_epi<NT, W_IN_CONTIG, true>(a, epilogue, grid, s) : launch_rows_epi<NT, W_IN_CONTIG, false>(a, epilogue, grid, s);
    if (layout == W_OUT_CONTIG) return full ? launch_rows_epi<NT, W_OUT_CONTIG, true>(a, epilogue, grid, s) : launch_rows_epi<NT, W_OUT_CONTIG, false>(a, epilogue, grid, s);
    return launch_rows_epi<NT, W_ANY, false>(a, epilogue, grid, s);
}

}  // namespace
}  // namespace envidr

using namespace envidr;

extern "C" {

int envidr_linear_rows(const float* x, uint32_t ldx, uint32_t M, uint32_t K, const float* W, int64_t w_stride_out, int64_t w_stride_in, uint32_t N,
                       const float* bias, const float* act, uint32_t ldact, int epilogue, float* y, uint32_t ldy, envidr_stream_t stream) {
    ENVIDR_REQUIRE(epilogue >= ENVIDR_ROWS_PLAIN && epilogue <= ENVIDR_ROWS_RELU_MASK, "linear_rows: unknown epilogue %d", epilogue);
    ENVIDR_REQUIRE(K >= 4 && K % 4 == 0 && N >= 1, "linear_rows: K=%u must be a positive multiple of 4, N=%u positive", K, N);
    if (M == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(x && W && y, "linear_rows: null pointer");
    ENVIDR_REQUIRE(ldx >= K && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "linear_rows: x rows must be 16-byte aligned (ldx=%u)", ldx);
    ENVIDR_REQUIRE(ldy >= N, "linear_rows: ldy=%u < N=%u", ldy, N);
    ENVIDR_REQUIRE((epilogue != ENVIDR_ROWS_BIAS && epilogue != ENVIDR_ROWS_BIAS_RELU) || bias, "linear_rows: this epilogue needs a bias");
    ENVIDR_REQUIRE(epilogue != ENVIDR_ROWS_RELU_MASK || (act && ldact >= N), "linear_rows: the mask epilogue needs act [M][>= N]");
    ENVIDR_REQUIRE(w_stride_out >= 0 && w_stride_in >= 0, "linear_rows: negative weight stride");
    const bool w_aligned = (reinterpret_cast<uintptr_t>(W) & 15) == 0;
    int layout = W_ANY;
    if (w_stride_in == 1 && w_stride_out % 4 == 0 && w_aligned) layout = W_IN_CONTIG;
    else if (w_stride_out == 1 && w_stride_in % 4 == 0 && N % 4 == 0 && w_aligned) layout = W_OUT_CONTIG;
    RowsArgs a{x, W, bias, act, y, w_stride_out, w_stride_in, ldx, ldact, ldy, M, K, N};
    hipStream_t s = as_stream(stream);
    const uint32_t row_blocks = ceil_div(M, 128u);
    if (N <= 32) return launch_rows_layout<1>(a, layout, epilogue, dim3(row_blocks, 1), s);
    if (N <= 64) return launch_rows_layout<2>(a, layout, epilogue, dim3(row_blocks, 1), s);
    if (N <= 128) return launch_rows_layout<4>(a, layout, epilogue, dim3(row_blocks, 1), s);
    // wider layers: column blocks of 128 or 64, whichever pads N less (ties: 128).  Blocks of 256 (eight tiles per wave) were slower at every
    // width measured (146 k rows: 256 x 256 0.22 -> 0.20 ms, 72 -> 256 0.088 -> 0.077, 160 x 160 0.13 -> 0.10 with 64-column blocks): half the work
    // per workgroup spreads 4.45 workgroups per CU more evenly, and 2 .. 3 workgroups fit a CU instead of 2.
    const uint32_t pad128 = ceil_div(N, 128u) * 128u, pad64 = ceil_div(N, 64u) * 64u;
    if (pad64 < pad128) return launch_rows_layout<2>(a, layout, epilogue, dim3(row_blocks, pad64 / 64u), s);
    return launch_rows_layout<4>(a, layout, epilogue, dim3(row_blocks, pad128 / 128u), s);
}

}  // extern "C"
