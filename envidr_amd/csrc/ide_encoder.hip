// Integrated directional encoding as a HIP operator -- the reference implements it in PyTorch
// (ide_encoder/ide_encoder.py:98-130: complex pow + a [.,17]x[17,36] matmul + exp).  One lane per
// direction, output row [Re(all terms) | Im(all terms)].
#include "sh_core.hip.h"
#include "rowio.hip.h"

using namespace envidr;

// One lane per direction, one wave per workgroup; the wave's 64 x 2N tile of outputs leaves through LDS as 16-byte stores
// (rowio.hip.h): written from the lanes, each of the 2N store instructions would touch 64 different cache lines.
template <int DEG_VIEW, bool ALIGNED>
__global__ void __launch_bounds__(64) k_ide_forward(const float* __restrict__ dirs, const float* __restrict__ roughness, float roughness_scalar,
                                                    uint32_t B, float* __restrict__ outputs) {
    constexpr int N = ide_terms(DEG_VIEW);
    __shared__ float s_tile[wave_tile_floats<2 * N>()];
    const uint32_t lane = threadIdx.x, row0 = blockIdx.x * 64u;
    const uint32_t b = min(row0 + lane, B - 1);
    const float kinv = roughness ? roughness[b] : roughness_scalar;
    float v[2 * N];
    ide_eval<DEG_VIEW>(dirs[3 * (size_t)b], dirs[3 * (size_t)b + 1], dirs[3 * (size_t)b + 2], kinv,
                       [&](int j, float re, float im) { v[j] = re; v[N + j] = im; });
    wave_store_rows<2 * N, ALIGNED>(s_tile, v, outputs + (size_t)row0 * 2 * N, min(64u, B - row0), lane);
}

extern "C" int envidr_ide_encode_forward(const float* dirs, const float* roughness_ptr, float roughness_scalar, uint32_t B,
                                         uint32_t deg_view, float* outputs, envidr_stream_t stream) {
    ENVIDR_REQUIRE(deg_view >= 1 && deg_view <= 5,
                   "ide_encode_forward: deg_view must be in [1, 5] (the reference raises ValueError above 5)");
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(dirs && outputs, "ide_encode_forward: null pointer");
    const dim3 grid(ceil_div(B, 64)), block(64);
    hipStream_t s = as_stream(stream);
    const bool al = aligned16(outputs);
#define ENVIDR_IDE(DEG)                                                                                                             \
    case DEG:                                                                                                                       \
        if (al) hipLaunchKernelGGL((k_ide_forward<DEG, true>), grid, block, 0, s, dirs, roughness_ptr, roughness_scalar, B, outputs); \
        else hipLaunchKernelGGL((k_ide_forward<DEG, false>), grid, block, 0, s, dirs, roughness_ptr, roughness_scalar, B, outputs);   \
        break;
    switch (deg_view) { ENVIDR_IDE(1) ENVIDR_IDE(2) ENVIDR_IDE(3) ENVIDR_IDE(4) ENVIDR_IDE(5) }
#undef ENVIDR_IDE
    return check_launch("k_ide_forward");
}

// backward: one lane per direction (reference: torch autograd through ide_encoder.py:98-130)
template <int DEG_VIEW>
__global__ void __launch_bounds__(kBlock) k_ide_backward(const float* __restrict__ grad, const float* __restrict__ dirs,
                                                         const float* __restrict__ roughness, float roughness_scalar, uint32_t B,
                                                         float* __restrict__ grad_dirs, float* __restrict__ grad_roughness) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    constexpr int N = ide_terms(DEG_VIEW);
    const float kinv = roughness ? roughness[b] : roughness_scalar;
    const float* g = grad + (size_t)b * 2 * N;
    float gd[3], gk;
    ide_grad<DEG_VIEW>(dirs[3 * (size_t)b], dirs[3 * (size_t)b + 1], dirs[3 * (size_t)b + 2], kinv,
                       [&](int j, float& gre, float& gim) { gre = g[j]; gim = g[N + j]; }, gd, gk);
    if (grad_dirs) { grad_dirs[3 * (size_t)b] = gd[0]; grad_dirs[3 * (size_t)b + 1] = gd[1]; grad_dirs[3 * (size_t)b + 2] = gd[2]; }
    if (grad_roughness) grad_roughness[b] = gk;
}

extern "C" int envidr_ide_encode_backward(const float* grad, const float* dirs, const float* roughness_ptr, float roughness_scalar,
                                          uint32_t B, uint32_t deg_view, float* grad_dirs, float* grad_roughness, envidr_stream_t stream) {
    ENVIDR_REQUIRE(deg_view >= 1 && deg_view <= 5, "ide_encode_backward: deg_view must be in [1, 5]");
    if (B == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(grad && dirs && (grad_dirs || grad_roughness), "ide_encode_backward: null pointer");
    const dim3 grid(ceil_div(B, kBlock)), block(kBlock);
    hipStream_t s = as_stream(stream);
    switch (deg_view) {
        case 1: hipLaunchKernelGGL(k_ide_backward<1>, grid, block, 0, s, grad, dirs, roughness_ptr, roughness_scalar, B, grad_dirs, grad_roughness); break;
        case 2: hipLaunchKernelGGL(k_ide_backward<2>, grid, block, 0, s, grad, dirs, roughness_ptr, roughness_scalar, B, grad_dirs, grad_roughness); break;
        case 3: hipLaunchKernelGGL(k_ide_backward<3>, grid, block, 0, s, grad, dirs, roughness_ptr, roughness_scalar, B, grad_dirs, grad_roughness); break;
        case 4: hipLaunchKernelGGL(k_ide_backward<4>, grid, block, 0, s, grad, dirs, roughness_ptr, roughness_scalar, B, grad_dirs, grad_roughness); break;
        case 5: hipLaunchKernelGGL(k_ide_backward<5>, grid, block, 0, s, grad, dirs, roughness_ptr, roughness_scalar, B, grad_dirs, grad_roughness); break;
    }
    return check_launch("k_ide_backward");
}
