/*
 * TEST INFRASTRUCTURE (oracle/_ref builder input) -- see kernel_keywords.h.
 *
 * C entry points that drive the reference's own kernel bodies on the CPU.  The bodies are NOT in
 * this repository: build_ref.py slices them out of /root/reference/<ext>/src/<ext>.cu into a
 * temporary directory at build time and passes that directory with -I; the five `#include
 * "..._kernels.inc"` lines below pull them in, each inside its own namespace because the
 * reference re-uses names (kernel_grid, div_round_up, atomicAdd...) across extensions.
 *
 * Entry-point signatures equal include/envidr_amd.h minus the stream argument and with host
 * pointers, so the tests bind the HIP library, the C restatement (oracle/c) and this library
 * through one table.  Launch geometry follows the reference's host functions.
 */
#ifndef ENVIDR_REF_KEYWORDS
#define ENVIDR_REF_KEYWORDS "kernel_keywords.h"          /* the CPU build; build_ref.py's device build passes "device_keywords.h" */
#endif
#include ENVIDR_REF_KEYWORDS

namespace ref_rm {
using ::atomicAdd;
#include "raymarching_kernels.inc"
}
namespace ref_hash {
using ::atomicAdd;
#include "hashencoder_kernels.inc"
}
namespace ref_grid {
using ::atomicAdd;
#include "gridencoder_kernels.inc"
}
namespace ref_freq {
#include "freqencoder_kernels.inc"
}
namespace ref_sh {
#include "shencoder_kernels.inc"
}

#define EXPORT extern "C" __attribute__((visibility("default")))

/* ---- raymarching (reference host functions raymarching.cu:148,201,230,261,292,326,511,704,824,947,1049) ---- */
EXPORT int ref_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                                  float min_near, float* nears, float* fars) {
    emu_launch(emu_blocks(N, 128), 1, 128, EMU_LAMBDA { ref_rm::kernel_near_far_from_aabb<float>(rays_o, rays_d, aabb, N, min_near, nears, fars); });
    return 0;
}
EXPORT int ref_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords) {
    emu_launch(emu_blocks(N, 128), 1, 128, EMU_LAMBDA { ref_rm::kernel_sph_from_ray<float>(rays_o, rays_d, radius, N, coords); });
    return 0;
}
EXPORT int ref_morton3D(const int32_t* coords, uint32_t N, int32_t* indices) {
    emu_launch(emu_blocks(N, 128), 1, 128, EMU_LAMBDA { ref_rm::kernel_morton3D(coords, N, indices); });
    return 0;
}
EXPORT int ref_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords) {
    emu_launch(emu_blocks(N, 128), 1, 128, EMU_LAMBDA { ref_rm::kernel_morton3D_invert(indices, N, coords); });
    return 0;
}
EXPORT int ref_packbits(const float* grid, uint32_t N, float thresh, uint8_t* bitfield) {
    emu_launch(emu_blocks(N, 128), 1, 128, EMU_LAMBDA { ref_rm::kernel_packbits<float>(grid, N, thresh, bitfield); });
    return 0;
}
EXPORT int ref_get_scatter_idx(const int32_t* rays, uint32_t N, int32_t* idx_map) {
    emu_launch(emu_blocks(N, 128), 1, 128, EMU_LAMBDA { ref_rm::kernel_get_scatter_idx(rays, N, idx_map); });
    return 0;
}
EXPORT int ref_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                                float dt_gamma, uint32_t max_steps, uint32_t early_stop_steps, uint32_t N, uint32_t C,
                                uint32_t H, uint32_t M, const float* nears, const float* fars, float* xyzs, float* dirs,
                                float* deltas, int32_t* rays, int32_t* counter, const float* noises) {
    emu_launch(emu_blocks(N, 128), 1, 128, EMU_LAMBDA {
        ref_rm::kernel_march_rays_train<float>(rays_o, rays_d, grid, bound, dt_gamma, max_steps, early_stop_steps, N, C, H,
                                               M, nears, fars, xyzs, dirs, deltas, rays, counter, noises);
    });
    return 0;
}
EXPORT int ref_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas,
                                            const int32_t* rays, uint32_t M, uint32_t N, float T_thresh,
                                            uint32_t accum_deltas, uint32_t input_alpha, float* weights_sum,
                                            float* depth, float* image, float* weights) {
    if (weights)
        emu_launch(emu_blocks(N, 128), 1, 128, EMU_LAMBDA {
            ref_rm::kernel_composite_rays_train_forward_with_weight<float>(sigmas, rgbs, deltas, rays, M, N, T_thresh, accum_deltas,
                                                                           input_alpha, weights_sum, depth, image, weights);
        });
    else
        emu_launch(emu_blocks(N, 128), 1, 128, EMU_LAMBDA {
            ref_rm::kernel_composite_rays_train_forward<float>(sigmas, rgbs, deltas, rays, M, N, T_thresh, accum_deltas,
                                                               input_alpha, weights_sum, depth, image);
        });
    return 0;
}
EXPORT int ref_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image,
                                             const float* grad_depth, const float* sigmas, const float* rgbs,
                                             const float* deltas, const int32_t* rays, const float* weights_sum,
                                             const float* image, const float* depth, uint32_t M, uint32_t N,
                                             float T_thresh, float* grad_sigmas, float* grad_rgbs, uint32_t accum_deltas,
                                             uint32_t input_alpha) {
    emu_launch(emu_blocks(N, 128), 1, 128, EMU_LAMBDA {
        ref_rm::kernel_composite_rays_train_backward<float>(grad_weights_sum, grad_image, grad_depth, sigmas, rgbs, deltas, rays,
                                                            weights_sum, image, depth, M, N, T_thresh, grad_sigmas, grad_rgbs,
                                                            accum_deltas, input_alpha);
    });
    return 0;
}
EXPORT int ref_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                          const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                          uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars, float* xyzs,
                          float* dirs, float* deltas, const float* noises) {
    emu_launch(emu_blocks(n_alive, 128), 1, 128, EMU_LAMBDA {
        ref_rm::kernel_march_rays<float>(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H,
                                         grid, nears, fars, xyzs, dirs, deltas, noises);
    });
    return 0;
}
EXPORT int ref_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, uint32_t accum_deltas,
                              uint32_t input_alpha, int32_t* rays_alive, float* rays_t, const float* sigmas,
                              const float* rgbs, const float* deltas, float* weights_sum, float* depth, float* image) {
    emu_launch(emu_blocks(n_alive, 128), 1, 128, EMU_LAMBDA {
        ref_rm::kernel_composite_rays<float>(n_alive, n_step, T_thresh, accum_deltas, input_alpha, rays_alive, rays_t, sigmas,
                                             rgbs, deltas, weights_sum, depth, image);
    });
    return 0;
}

/* ---- hashencoder (reference wrappers hashencoder.cu:598-720) ---- */
template <uint32_t D, uint32_t C>
static void hash_fwd(const float* in, const float* emb, const int* off, float* out, uint32_t B, uint32_t L, float S, uint32_t H,
                     bool g, float* dy_dx) {
    emu_launch(emu_blocks(B, 512), L, 512, EMU_LAMBDA { ref_hash::kernel_grid<float, D, C>(in, emb, off, out, B, L, S, H, g, dy_dx); });
}
template <uint32_t D, uint32_t C, uint32_t NC>
static void hash_bwd(const float* grad, const float* in, const float* emb, const int* off, float* gemb, uint32_t B, uint32_t L,
                     float S, uint32_t H, bool g, const float* dy_dx, float* gin) {
    if (gemb)
        emu_launch(emu_blocks(B * C / NC, 256), L, 256,
                   EMU_LAMBDA { ref_hash::kernel_grid_backward<float, D, C, NC>(grad, in, emb, off, gemb, B, L, S, H); });
    if (g) emu_launch(emu_blocks(B * D, 256), 1, 256, EMU_LAMBDA { ref_hash::kernel_input_backward<float, D, C>(grad, dy_dx, gin, B, L); });
}
template <uint32_t D, uint32_t C, uint32_t NC>
static void hash_bwd2(const float* grad, const float* in, const float* emb, const int* off, uint32_t B, uint32_t L, float S,
                      uint32_t H, const float* dy_dx, const float* ggi, float* gg, float* g2e) {
    emu_launch(emu_blocks(B * C / NC, 256), L, 256,
               EMU_LAMBDA { ref_hash::kernel_grid_second_backward_grad<float, D, C, NC>(grad, in, emb, off, ggi, dy_dx, gg, B, L, S, H); });
    emu_launch(emu_blocks(B * C / NC, 256), L, 256, EMU_LAMBDA {
        ref_hash::kernel_grid_second_backward_embedding<float, D, C, NC>(grad, in, emb, off, ggi, dy_dx, g2e, B, L, S, H);
    });
}
#define DC_SWITCH(CALL2, CALL3)                                                  \
    if (D == 2) { switch (C) { case 1: CALL2(1, 1); break; case 2: CALL2(2, 2); break; case 4: CALL2(4, 2); break; case 8: CALL2(8, 2); break; default: return -1; } } \
    else if (D == 3) { switch (C) { case 1: CALL3(1, 1); break; case 2: CALL3(2, 2); break; case 4: CALL3(4, 2); break; case 8: CALL3(8, 2); break; default: return -1; } } \
    else return -1;

EXPORT int ref_hash_encode_forward(const float* inputs, const float* embeddings, const int32_t* offsets, float* outputs,
                                   uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                   int calc_grad_inputs, float* dy_dx) {
#define F2(c, nc) hash_fwd<2, c>(inputs, embeddings, offsets, outputs, B, L, S, H, calc_grad_inputs != 0, dy_dx)
#define F3(c, nc) hash_fwd<3, c>(inputs, embeddings, offsets, outputs, B, L, S, H, calc_grad_inputs != 0, dy_dx)
    DC_SWITCH(F2, F3)
#undef F2
#undef F3
    return 0;
}
EXPORT int ref_hash_encode_backward(const float* grad, const float* inputs, const float* embeddings,
                                    const int32_t* offsets, float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                    uint32_t L, float S, uint32_t H, int calc_grad_inputs, const float* dy_dx,
                                    float* grad_inputs) {
#define F2(c, nc) hash_bwd<2, c, nc>(grad, inputs, embeddings, offsets, grad_embeddings, B, L, S, H, calc_grad_inputs != 0, dy_dx, grad_inputs)
#define F3(c, nc) hash_bwd<3, c, nc>(grad, inputs, embeddings, offsets, grad_embeddings, B, L, S, H, calc_grad_inputs != 0, dy_dx, grad_inputs)
    DC_SWITCH(F2, F3)
#undef F2
#undef F3
    return 0;
}
EXPORT int ref_hash_encode_second_backward(const float* grad, const float* inputs, const float* embeddings,
                                           const int32_t* offsets, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                           uint32_t H, int calc_grad_inputs, const float* dy_dx,
                                           const float* grad_grad_inputs, float* grad_grad, float* grad2_embeddings) {
    (void)calc_grad_inputs;
    if (C == 1) return -1;
#define F2(c, nc) hash_bwd2<2, c, nc>(grad, inputs, embeddings, offsets, B, L, S, H, dy_dx, grad_grad_inputs, grad_grad, grad2_embeddings)
#define F3(c, nc) hash_bwd2<3, c, nc>(grad, inputs, embeddings, offsets, B, L, S, H, dy_dx, grad_grad_inputs, grad_grad, grad2_embeddings)
    DC_SWITCH(F2, F3)
#undef F2
#undef F3
    return 0;
}

/* ---- gridencoder (reference wrappers gridencoder.cu:345-420) ---- */
template <uint32_t D, uint32_t C>
static void grid_fwd(const float* in, const float* emb, const int* off, float* out, uint32_t B, uint32_t L, float S, uint32_t H,
                     float* dy_dx, uint32_t gt, bool ac) {
    emu_launch(emu_blocks(B, 512), L, 512, EMU_LAMBDA { ref_grid::kernel_grid<float, D, C>(in, emb, off, out, B, L, S, H, dy_dx, gt, ac); });
}
template <uint32_t D, uint32_t C, uint32_t NC>
static void grid_bwd(const float* grad, const float* in, const float* emb, const int* off, float* gemb, uint32_t B, uint32_t L,
                     float S, uint32_t H, const float* dy_dx, float* gin, uint32_t gt, bool ac) {
    emu_launch(emu_blocks(B * C / NC, 256), L, 256,
               EMU_LAMBDA { ref_grid::kernel_grid_backward<float, D, C, NC>(grad, in, emb, off, gemb, B, L, S, H, gt, ac); });
    if (dy_dx) emu_launch(emu_blocks(B * D, 256), 1, 256, EMU_LAMBDA { ref_grid::kernel_input_backward<float, D, C>(grad, dy_dx, gin, B, L); });
}
#define GRID_DISPATCH(FN)                                                                                          \
    switch (D * 10 + C) {                                                                                          \
        FN(1, 1, 1) FN(1, 2, 2) FN(1, 4, 2) FN(1, 8, 2) FN(2, 1, 1) FN(2, 2, 2) FN(2, 4, 2) FN(2, 8, 2)             \
        FN(3, 1, 1) FN(3, 2, 2) FN(3, 4, 2) FN(3, 8, 2) FN(4, 1, 1) FN(4, 2, 2) FN(4, 4, 2) FN(4, 8, 2)             \
        FN(5, 1, 1) FN(5, 2, 2) FN(5, 4, 2) FN(5, 8, 2)                                                             \
        default: return -1;                                                                                        \
    }
EXPORT int ref_grid_encode_forward(const float* inputs, const float* embeddings, const int32_t* offsets, float* outputs,
                                   uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, float* dy_dx,
                                   uint32_t gridtype, int align_corners) {
#define FN(d, c, nc) case d * 10 + c: grid_fwd<d, c>(inputs, embeddings, offsets, outputs, B, L, S, H, dy_dx, gridtype, align_corners != 0); break;
    GRID_DISPATCH(FN)
#undef FN
    return 0;
}
EXPORT int ref_grid_encode_backward(const float* grad, const float* inputs, const float* embeddings,
                                    const int32_t* offsets, float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                    uint32_t L, float S, uint32_t H, const float* dy_dx, float* grad_inputs,
                                    uint32_t gridtype, int align_corners) {
#define FN(d, c, nc) case d * 10 + c: grid_bwd<d, c, nc>(grad, inputs, embeddings, offsets, grad_embeddings, B, L, S, H, dy_dx, grad_inputs, gridtype, align_corners != 0); break;
    GRID_DISPATCH(FN)
#undef FN
    return 0;
}

/* ---- the at::Half instantiations of the same kernel templates (AT_DISPATCH_FLOATING_TYPES_AND_HALF: hashencoder.cu:747,778,
 *      gridencoder.cu:443,474).  Pointers to 16-bit storage; hashencoder narrows inputs, table, outputs and dy_dx, gridencoder
 *      keeps the inputs in float (its kernels take `const float* inputs`). ---- */
typedef at::Half H16;
template <uint32_t D, uint32_t C>
static void hash_fwd_h(const H16* in, const H16* emb, const int* off, H16* out, uint32_t B, uint32_t L, float S, uint32_t H, bool g, H16* dy_dx) {
    emu_launch(emu_blocks(B, 512), L, 512, EMU_LAMBDA { ref_hash::kernel_grid<H16, D, C>(in, emb, off, out, B, L, S, H, g, dy_dx); });
}
template <uint32_t D, uint32_t C, uint32_t NC>
static void hash_bwd_h(const H16* grad, const H16* in, const H16* emb, const int* off, H16* gemb, uint32_t B, uint32_t L, float S, uint32_t H,
                       bool g, const H16* dy_dx, H16* gin) {
    if (gemb)
        emu_launch(emu_blocks(B * C / NC, 256), L, 256, EMU_LAMBDA { ref_hash::kernel_grid_backward<H16, D, C, NC>(grad, in, emb, off, gemb, B, L, S, H); });
    if (g) emu_launch(emu_blocks(B * D, 256), 1, 256, EMU_LAMBDA { ref_hash::kernel_input_backward<H16, D, C>(grad, dy_dx, gin, B, L); });
}
EXPORT int ref_hash_encode_forward_f16(const uint16_t* inputs, const uint16_t* embeddings, const int32_t* offsets, uint16_t* outputs,
                                       uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, uint16_t* dy_dx) {
#define F2(c, nc) hash_fwd_h<2, c>((const H16*)inputs, (const H16*)embeddings, offsets, (H16*)outputs, B, L, S, H, calc_grad_inputs != 0, (H16*)dy_dx)
#define F3(c, nc) hash_fwd_h<3, c>((const H16*)inputs, (const H16*)embeddings, offsets, (H16*)outputs, B, L, S, H, calc_grad_inputs != 0, (H16*)dy_dx)
    DC_SWITCH(F2, F3)
#undef F2
#undef F3
    return 0;
}
EXPORT int ref_hash_encode_backward_f16(const uint16_t* grad, const uint16_t* inputs, const uint16_t* embeddings, const int32_t* offsets,
                                        uint16_t* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                        int calc_grad_inputs, const uint16_t* dy_dx, uint16_t* grad_inputs) {
#define F2(c, nc) hash_bwd_h<2, c, nc>((const H16*)grad, (const H16*)inputs, (const H16*)embeddings, offsets, (H16*)grad_embeddings, B, L, S, H, calc_grad_inputs != 0, (const H16*)dy_dx, (H16*)grad_inputs)
#define F3(c, nc) hash_bwd_h<3, c, nc>((const H16*)grad, (const H16*)inputs, (const H16*)embeddings, offsets, (H16*)grad_embeddings, B, L, S, H, calc_grad_inputs != 0, (const H16*)dy_dx, (H16*)grad_inputs)
    DC_SWITCH(F2, F3)
#undef F2
#undef F3
    return 0;
}
template <uint32_t D, uint32_t C, uint32_t NC>
static void hash_bwd2_h(const H16* grad, const H16* in, const H16* emb, const int* off, uint32_t B, uint32_t L, float S, uint32_t H,
                        const H16* dy_dx, const H16* ggi, H16* gg, H16* g2e) {
    emu_launch(emu_blocks(B * C / NC, 256), L, 256,
               EMU_LAMBDA { ref_hash::kernel_grid_second_backward_grad<H16, D, C, NC>(grad, in, emb, off, ggi, dy_dx, gg, B, L, S, H); });
    emu_launch(emu_blocks(B * C / NC, 256), L, 256, EMU_LAMBDA {
        ref_hash::kernel_grid_second_backward_embedding<H16, D, C, NC>(grad, in, emb, off, ggi, dy_dx, g2e, B, L, S, H);
    });
}
EXPORT int ref_hash_encode_second_backward_f16(const uint16_t* grad, const uint16_t* inputs, const uint16_t* embeddings, const int32_t* offsets,
                                               uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs,
                                               const uint16_t* dy_dx, const uint16_t* grad_grad_inputs, uint16_t* grad_grad, uint16_t* grad2_embeddings) {
    (void)calc_grad_inputs;
    if (C == 1) return -1;
#define F2(c, nc) hash_bwd2_h<2, c, nc>((const H16*)grad, (const H16*)inputs, (const H16*)embeddings, offsets, B, L, S, H, (const H16*)dy_dx, (const H16*)grad_grad_inputs, (H16*)grad_grad, (H16*)grad2_embeddings)
#define F3(c, nc) hash_bwd2_h<3, c, nc>((const H16*)grad, (const H16*)inputs, (const H16*)embeddings, offsets, B, L, S, H, (const H16*)dy_dx, (const H16*)grad_grad_inputs, (H16*)grad_grad, (H16*)grad2_embeddings)
    DC_SWITCH(F2, F3)
#undef F2
#undef F3
    return 0;
}
template <uint32_t D, uint32_t C>
static void grid_fwd_h(const float* in, const H16* emb, const int* off, H16* out, uint32_t B, uint32_t L, float S, uint32_t H, H16* dy_dx,
                       uint32_t gt, bool ac) {
    emu_launch(emu_blocks(B, 512), L, 512, EMU_LAMBDA { ref_grid::kernel_grid<H16, D, C>(in, emb, off, out, B, L, S, H, dy_dx, gt, ac); });
}
template <uint32_t D, uint32_t C, uint32_t NC>
static void grid_bwd_h(const H16* grad, const float* in, const H16* emb, const int* off, H16* gemb, uint32_t B, uint32_t L, float S, uint32_t H,
                       const H16* dy_dx, H16* gin, uint32_t gt, bool ac) {
    emu_launch(emu_blocks(B * C / NC, 256), L, 256, EMU_LAMBDA { ref_grid::kernel_grid_backward<H16, D, C, NC>(grad, in, emb, off, gemb, B, L, S, H, gt, ac); });
    if (dy_dx) emu_launch(emu_blocks(B * D, 256), 1, 256, EMU_LAMBDA { ref_grid::kernel_input_backward<H16, D, C>(grad, dy_dx, gin, B, L); });
}
EXPORT int ref_grid_encode_forward_f16(const float* inputs, const uint16_t* embeddings, const int32_t* offsets, uint16_t* outputs, uint32_t B,
                                       uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint16_t* dy_dx, uint32_t gridtype, int align_corners) {
#define FN(d, c, nc) case d * 10 + c: grid_fwd_h<d, c>(inputs, (const H16*)embeddings, offsets, (H16*)outputs, B, L, S, H, (H16*)dy_dx, gridtype, align_corners != 0); break;
    GRID_DISPATCH(FN)
#undef FN
    return 0;
}
EXPORT int ref_grid_encode_backward_f16(const uint16_t* grad, const float* inputs, const uint16_t* embeddings, const int32_t* offsets,
                                        uint16_t* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                        const uint16_t* dy_dx, uint16_t* grad_inputs, uint32_t gridtype, int align_corners) {
#define FN(d, c, nc) case d * 10 + c: grid_bwd_h<d, c, nc>((const H16*)grad, inputs, (const H16*)embeddings, offsets, (H16*)grad_embeddings, B, L, S, H, (const H16*)dy_dx, (H16*)grad_inputs, gridtype, align_corners != 0); break;
    GRID_DISPATCH(FN)
#undef FN
    return 0;
}

/* ---- freqencoder (freqencoder.cu:97,113) ---- */
EXPORT int ref_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs) {
    emu_launch(emu_blocks(B * C, 128), 1, 128, EMU_LAMBDA { ref_freq::kernel_freq(inputs, B, D, deg, C, outputs); });
    return 0;
}
EXPORT int ref_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                                    float* grad_inputs) {
    emu_launch(emu_blocks(B * D, 128), 1, 128, EMU_LAMBDA { ref_freq::kernel_freq_backward(grad, outputs, B, D, deg, C, grad_inputs); });
    return 0;
}

/* ---- shencoder (shencoder.cu:387,395) ---- */
EXPORT int ref_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C, float* dy_dx) {
    emu_launch(emu_blocks(B, 256), 1, 256, EMU_LAMBDA { ref_sh::kernel_sh<float>(inputs, outputs, B, D, C, dy_dx); });
    return 0;
}
EXPORT int ref_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C,
                                  const float* dy_dx, float* grad_inputs) {
    emu_launch(emu_blocks(B * D, 256), 1, 256, EMU_LAMBDA { ref_sh::kernel_sh_backward<float>(grad, inputs, B, D, C, dy_dx, grad_inputs); });
    return 0;
}

/* ---- shencoder on at::Half (the dispatch of shencoder.cu:413,435) ---- */
EXPORT int ref_sh_encode_forward_f16(const uint16_t* inputs, uint16_t* outputs, uint32_t B, uint32_t D, uint32_t C, uint16_t* dy_dx) {
    emu_launch(emu_blocks(B, 256), 1, 256, EMU_LAMBDA { ref_sh::kernel_sh<H16>((const H16*)inputs, (H16*)outputs, B, D, C, (H16*)dy_dx); });
    return 0;
}
EXPORT int ref_sh_encode_backward_f16(const uint16_t* grad, const uint16_t* inputs, uint32_t B, uint32_t D, uint32_t C,
                                      const uint16_t* dy_dx, uint16_t* grad_inputs) {
    emu_launch(emu_blocks(B * D, 256), 1, 256,
               EMU_LAMBDA { ref_sh::kernel_sh_backward<H16>((const H16*)grad, (const H16*)inputs, B, D, C, (const H16*)dy_dx, (H16*)grad_inputs); });
    return 0;
}
