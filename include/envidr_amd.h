/*
 * envidr_amd -- C-ABI of the MI355X-native ENVIDR render hot path.
 *
 * Every entry point below replaces one free function the reference exports from its five
 * pybind11 extension modules (the `_backend.*` calls made by the Python autograd wrappers).
 * The reference passes `at::Tensor`; this ABI passes what those tensors are underneath:
 * a device pointer to contiguous memory, plus the same scalar arguments in the same order.
 * The trailing `stream` is the hipStream_t (as void*) the work is enqueued on; the reference
 * always used the default stream.
 *
 * Conventions
 *   - all pointers are DEVICE pointers into HBM unless a parameter is named *_host;
 *   - all tensors are contiguous row-major fp32 / int32 / uint8 exactly as the reference lays
 *     them out; outputs are caller-allocated and written in place (reference contract);
 *   - every function is asynchronous w.r.t. the host and returns 0 on success or a negative
 *     ENVIDR_E* code (nothing is enqueued on error); envidr_last_error() gives the text;
 *   - a count of zero elements is a successful no-op (the reference would launch a zero grid).
 *
 * This header is plain C; no torch, HIP or C++ types appear in any signature.
 */
#ifndef ENVIDR_AMD_H
#define ENVIDR_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ENVIDR_OK            0
#define ENVIDR_EINVAL       (-1)   /* bad argument (null pointer, unsupported D/C/degree ...) */
#define ENVIDR_ELAUNCH      (-2)   /* HIP reported a launch / runtime error                   */

typedef void* envidr_stream_t;     /* hipStream_t */

const char* envidr_last_error(void);
/* ABI version of this library: bumped on any signature or struct-layout change and when entry points are added (see csrc/capi.hip for the history). */
int envidr_abi_version(void);
/* Releases what the library keeps between calls outside the caller's allocator: the range-mask scratch of the table-gradient scatters
 * (hash_encode_backward / _second_backward with >= 32 k points: one buffer per (device, stream), sized to the largest batch seen, at most
 * 64 MiB).  Waits for the device; returns the number of bytes released.  No reference counterpart (torch's caching allocator there). */
uint64_t envidr_release_scratch(void);

/* ------------------------------------------------------------------------------------------
 * raymarching  (reference: raymarching/src/raymarching.h:7-18, bindings.cpp:5-19)
 * ------------------------------------------------------------------------------------------ */

/* raymarching.cu:148 near_far_from_aabb -- slab test of N rays against aabb[6]. */
int envidr_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb,
                              uint32_t N, float min_near, float* nears, float* fars,
                              envidr_stream_t stream);

/* utils.get_rays, full-image branch (nerf/utils.py:109-209, :193-207): pixel centres at +0.5 -> unit camera directions ->
 * rotated by the camera-to-world pose; origins = the pose's translation.  poses: device [B,4,4] row-major; outputs device
 * [B, H*W, 3], pixels row-major.  (The reference computes this with torch ops; here it is one launch.) */
int envidr_get_rays(const float* poses, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W, uint32_t B,
                    float* rays_o, float* rays_d, envidr_stream_t stream);

/* raymarching.cu:201 sph_from_ray -- far hit with sphere(radius) -> (theta,phi) in [-1,1]^2. */
int envidr_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N,
                        float* coords, envidr_stream_t stream);

/* raymarching.cu:230 morton3D / :261 morton3D_invert -- 10-bit-per-axis interleave. */
int envidr_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, envidr_stream_t stream);
int envidr_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords,
                           envidr_stream_t stream);

/* raymarching.cu:292 packbits -- N = number of output BYTES; bit i = grid[8n+i] > thresh. */
int envidr_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield,
                    envidr_stream_t stream);

/* raymarching.cu:326 get_scatter_idx -- rays[N,3]=(idx,offset,count) -> idx_map[offset..]. */
int envidr_get_scatter_idx(const int32_t* rays, uint32_t N, int32_t* idx_map,
                           envidr_stream_t stream);

/* raymarching.cu:511 march_rays_train -- two-pass training marcher with a global counter. */
int envidr_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid,
                            float bound, float dt_gamma, uint32_t max_steps,
                            uint32_t early_stop_steps, uint32_t N, uint32_t C, uint32_t H,
                            uint32_t M, const float* nears, const float* fars, float* xyzs,
                            float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                            const float* noises, envidr_stream_t stream);

/* raymarching.cu:704 composite_rays_train_forward -- weights may be NULL (reference: an empty
 * tensor selects the kernel without the per-sample weight output). */
int envidr_composite_rays_train_forward(const float* sigmas, const float* rgbs,
                                        const float* deltas, const int32_t* rays, uint32_t M,
                                        uint32_t N, float T_thresh, uint32_t accum_deltas,
                                        uint32_t input_alpha, float* weights_sum, float* depth,
                                        float* image, float* weights, envidr_stream_t stream);

/* raymarching.cu:824 composite_rays_train_backward. */
int envidr_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image,
                                         const float* grad_depth, const float* sigmas,
                                         const float* rgbs, const float* deltas,
                                         const int32_t* rays, const float* weights_sum,
                                         const float* image, const float* depth, uint32_t M,
                                         uint32_t N, float T_thresh, float* grad_sigmas,
                                         float* grad_rgbs, uint32_t accum_deltas,
                                         uint32_t input_alpha, envidr_stream_t stream);

/* raymarching.cu:947 march_rays -- inference marcher: up to n_step occupied samples for each of
 * the first n_alive ids of rays_alive, resuming at rays_t[id].  xyzs/dirs/deltas must be
 * zero-filled by the caller (reference wrapper allocates them with torch.zeros). */
int envidr_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive,
                      const float* rays_t, const float* rays_o, const float* rays_d,
                      float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                      const uint8_t* grid, const float* nears, const float* fars, float* xyzs,
                      float* dirs, float* deltas, const float* noises, envidr_stream_t stream);

/* raymarching.cu:1049 composite_rays -- inference compositor, in place on
 * weights_sum/depth/image/rays_t; writes -1 into rays_alive[n] for rays that stopped early. */
int envidr_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh,
                          uint32_t accum_deltas, uint32_t input_alpha, int32_t* rays_alive,
                          float* rays_t, const float* sigmas, const float* rgbs,
                          const float* deltas, float* weights_sum, float* depth, float* image,
                          envidr_stream_t stream);

/* NEW (no reference counterpart; replaces the host-syncing `rays_alive[rays_alive >= 0]` of
 * nerf/render_func/cuda_ray.py:345): order-preserving compaction of the non-negative ids of
 * rays_alive[0..n_alive) into out_alive, count written to *out_count (device int32). */
int envidr_compact_alive(uint32_t n_alive, const int32_t* rays_alive, int32_t* out_alive,
                         int32_t* out_count, envidr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * hashencoder  (reference: hashencoder/src/hashencoder.h:13-15)
 *   inputs [B,D] in [0,1]; embeddings [sum T_l, C]; offsets int32 [L+1] (device);
 *   outputs [L,B,C] (level-major); dy_dx [B, L*D*C] or NULL when !calc_grad_inputs.
 *   D in {2,3}; C in {1,2,4,8}; S = log2(per_level_scale); H = base resolution.
 * ------------------------------------------------------------------------------------------ */
int envidr_hash_encode_forward(const float* inputs, const float* embeddings,
                               const int32_t* offsets, float* outputs, uint32_t B, uint32_t D,
                               uint32_t C, uint32_t L, float S, uint32_t H,
                               int calc_grad_inputs, float* dy_dx, envidr_stream_t stream);

/* hashencoder.cu:762 -- grad [L,B,C]; accumulates (atomic) into grad_embeddings (caller
 * zero-fills); grad_inputs [B,D] written when calc_grad_inputs.  grad_embeddings may be NULL to
 * skip the table scatter (inference normals need only grad_inputs; the reference always pays for
 * a zeros_like(table) + scatter). */
int envidr_hash_encode_backward(const float* grad, const float* inputs, const float* embeddings,
                                const int32_t* offsets, float* grad_embeddings, uint32_t B,
                                uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                int calc_grad_inputs, const float* dy_dx, float* grad_inputs,
                                envidr_stream_t stream);

/* hashencoder.cu:795 -- double backward (eikonal loss through the normals). */
int envidr_hash_encode_second_backward(const float* grad, const float* inputs,
                                       const float* embeddings, const int32_t* offsets,
                                       uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                       uint32_t H, int calc_grad_inputs, const float* dy_dx,
                                       const float* grad_grad_inputs, float* grad_grad,
                                       float* grad2_embeddings, envidr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * gridencoder  (reference: gridencoder/src/gridencoder.h:12-13)
 *   as hashencoder but linear interpolation, +0.5 offset unless align_corners, stride res+1,
 *   gridtype 0 = hash / 1 = tiled, D in 1..5.  dy_dx / grad_inputs may be NULL.
 * ------------------------------------------------------------------------------------------ */
int envidr_grid_encode_forward(const float* inputs, const float* embeddings,
                               const int32_t* offsets, float* outputs, uint32_t B, uint32_t D,
                               uint32_t C, uint32_t L, float S, uint32_t H, float* dy_dx,
                               uint32_t gridtype, int align_corners, envidr_stream_t stream);
int envidr_grid_encode_backward(const float* grad, const float* inputs, const float* embeddings,
                                const int32_t* offsets, float* grad_embeddings, uint32_t B,
                                uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                const float* dy_dx, float* grad_inputs, uint32_t gridtype,
                                int align_corners, envidr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * freqencoder  (reference: freqencoder/src/freqencoder.h:6-9)   C = D + 2*D*deg
 * ------------------------------------------------------------------------------------------ */
int envidr_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg,
                               uint32_t C, float* outputs, envidr_stream_t stream);
int envidr_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D,
                                uint32_t deg, uint32_t C, float* grad_inputs,
                                envidr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * shencoder  (reference: shencoder/src/shencoder.h:9-10)   C = degree (1..8), outputs [B,C*C]
 *   dy_dx [B, 3*C*C] or NULL.  backward ACCUMULATES into grad_inputs (reference: `+=` into a
 *   zero-filled tensor, shencoder.cu:376).
 * ------------------------------------------------------------------------------------------ */
int envidr_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D,
                             uint32_t C, float* dy_dx, envidr_stream_t stream);
int envidr_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D,
                              uint32_t C, const float* dy_dx, float* grad_inputs,
                              envidr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ABI 6: half-precision tables -- the `scalar_t = at::Half` side of the reference's
 * AT_DISPATCH_FLOATING_TYPES_AND_HALF for the two grid encoders (hashencoder.cu:747,778; gridencoder.cu:443,474), i.e. what
 * `hashencoder/hashgrid.py:19` (custom_fwd(cast_inputs=torch.half)) and `gridencoder/grid.py:37-40` (half table under autocast)
 * dispatch to.  uint16_t* = IEEE binary16 storage (torch.half `data_ptr()`).  hashencoder narrows inputs, table, outputs,
 * dy_dx and gradients; gridencoder keeps fp32 inputs (its kernels take `const float* inputs`).  Arithmetic narrows where
 * c10::Half narrows (csrc/grid_half.hip); same argument order as the fp32 entry points.  hash_encode_second_backward_f16
 * (hashencoder.cu:817; all tensors fp16) completes the hash encoder; sh_encode_*_f16 below.  Not provided in half: freq (freqencoder.cu takes
 * `data_ptr<float>()`) and raymarching (raymarching.py `.float()`s every input; raymarching.cu:90 "scalar_t should always be
 * float in use").
 * ------------------------------------------------------------------------------------------ */
int envidr_hash_encode_forward_f16(const uint16_t* inputs, const uint16_t* embeddings, const int32_t* offsets,
                                   uint16_t* outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                   float S, uint32_t H, int calc_grad_inputs, uint16_t* dy_dx,
                                   envidr_stream_t stream);
int envidr_hash_encode_backward_f16(const uint16_t* grad, const uint16_t* inputs, const uint16_t* embeddings,
                                    const int32_t* offsets, uint16_t* grad_embeddings, uint32_t B, uint32_t D,
                                    uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs,
                                    const uint16_t* dy_dx, uint16_t* grad_inputs, envidr_stream_t stream);
int envidr_hash_encode_second_backward_f16(const uint16_t* grad, const uint16_t* inputs, const uint16_t* embeddings,
                                           const int32_t* offsets, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                           uint32_t H, int calc_grad_inputs, const uint16_t* dy_dx,
                                           const uint16_t* grad_grad_inputs, uint16_t* grad_grad,
                                           uint16_t* grad2_embeddings, envidr_stream_t stream);
/* shencoder.cu:413,435 on at::Half: half inputs / outputs / dy_dx / gradients.  The forward evaluates the basis in fp32 and rounds once
 * (within a few fp16 ulp of the reference's half instantiation, which rounds every monomial; not bit-identical); the backward is the
 * reference's Half arithmetic exactly.  The reference's own wrapper casts to float32 (sphere_harmonics.py:16) and never calls these. */
int envidr_sh_encode_forward_f16(const uint16_t* inputs, uint16_t* outputs, uint32_t B, uint32_t D, uint32_t C, uint16_t* dy_dx,
                                 envidr_stream_t stream);
int envidr_sh_encode_backward_f16(const uint16_t* grad, const uint16_t* inputs, uint32_t B, uint32_t D, uint32_t C,
                                  const uint16_t* dy_dx, uint16_t* grad_inputs, envidr_stream_t stream);
int envidr_grid_encode_forward_f16(const float* inputs, const uint16_t* embeddings, const int32_t* offsets,
                                   uint16_t* outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                   float S, uint32_t H, uint16_t* dy_dx, uint32_t gridtype, int align_corners,
                                   envidr_stream_t stream);
int envidr_grid_encode_backward_f16(const uint16_t* grad, const float* inputs, const uint16_t* embeddings,
                                    const int32_t* offsets, uint16_t* grad_embeddings, uint32_t B, uint32_t D,
                                    uint32_t C, uint32_t L, float S, uint32_t H, const uint16_t* dy_dx,
                                    uint16_t* grad_inputs, uint32_t gridtype, int align_corners,
                                    envidr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ide_encoder  (reference: ide_encoder/ide_encoder.py:98-130 -- pure PyTorch there; a HIP op
 * here).  dirs [B,3]; roughness: per-sample [B] when roughness_ptr != NULL, else the scalar
 * roughness_scalar; deg_view in 1..5; outputs [B, 2*(2^deg_view - 1 + deg_view)] = [Re | Im].
 * ------------------------------------------------------------------------------------------ */
int envidr_ide_encode_forward(const float* dirs, const float* roughness_ptr,
                              float roughness_scalar, uint32_t B, uint32_t deg_view,
                              float* outputs, envidr_stream_t stream);
/* ABI 6: its gradient -- what torch autograd computes through the reference's forward, which the training branch of
 * run_cuda differentiates (nerf/render_func/cuda_ray.py:118-119 -> renderer.py:147-180).  grad [B, 2 n]: upstream gradient of
 * the outputs; grad_dirs [B,3] and / or grad_roughness [B] (per-direction d / d kappa_inv; sum it for a shared scalar): either
 * may be NULL. */
int envidr_ide_encode_backward(const float* grad, const float* dirs, const float* roughness_ptr,
                               float roughness_scalar, uint32_t B, uint32_t deg_view,
                               float* grad_dirs, float* grad_roughness, envidr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ENVIDR_AMD_H */
