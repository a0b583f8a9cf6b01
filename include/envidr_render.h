/*
 * envidr_render -- C-ABI of the fused inference render path.
 *
 * This entry point replaces, as ONE persistent-wave kernel launch, the whole host-driven
 * march -> encode -> MLP -> composite -> compact loop of the reference's inference branch
 *     nerf/render_func/cuda_ray.py:238-359   (run_cuda, `else:` branch)
 * together with the per-sample model code it calls
 *     nerf/network.py:381-522   forward_geometry / forward_sigma   (hash grid -> SDF MLP -> density)
 *     nerf/renderer.py:182-198  compute_normal                      (analytic instead of autograd)
 *     nerf/renderer.py:147-180  get_color_mlp_extra_params          (reflection, IDE x2, n.v)
 *     nerf/network.py:524-698   forward_color                       (env MLP x2, diffuse, specular)
 * for the network family of configs/scenes/toaster.ini / configs/neural_renderer.ini
 * (hashgrid_diff position encoding, SDF + Laplace density, IDE-fed environment MLP, diffuse +
 * specular heads, sigmoid colours, unit-norm feature activations).
 *
 * A renderer built on the reference would bind it in place of run_cuda's loop.
 * Plain C: device pointers, host scalars, no torch / HIP types.
 */
#ifndef ENVIDR_RENDER_H
#define ENVIDR_RENDER_H

#include <stdint.h>
#include "envidr_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ENVIDR_MAX_LEVELS 16

/* ---- weight packing (host side, run once per model load) ------------------------------------
 * The fused kernel streams each layer's weights in a layout pre-permuted for the MFMA tiles
 * (envidr_amd/csrc/mlp_mfma.hip.h).  These helpers convert a torch-style row-major
 * nn.Linear.weight [out, in] (HOST pointers) into that layout.
 *   k_order: 0 = the layer's input is per-sample features written by scalar code ("lane order"),
 *            1 = the layer's input is the previous layer's output tiles       ("tile order"),
 *            2 = tile order for a layer of AT MOST 16 OUTPUTS (envidr_pack_layer only): the layer runs on
 *                16-row MFMA blocks, one 64-float fragment per reduction step (ABI 7; E4, D2, S3, renv R4).
 *            3 = tile order, 4 = lane order, each with the layer's LAST 16 reduction steps laid out tile-major (envidr_pack_layer
 *                only).  ABI 7: when T = env_hidden / 32 is even and >= 4 (256, 128) and the first layer has at least T + 18 reduction
 *                steps (IDE degree 5: 36; degree 4 has 19 and does not qualify at hidden 128) the environment pass stages a layer's
 *                first input tile inside the layer before it, and its blob must then be E1 (k_order 4) | E2 (3) | E3 (3) | E4 (2);
 *                otherwise (hidden 160; IDE degree 4 with hidden 128) E1 (0) | E2 (1) | E3 (1) | E4 (2).
 *   transpose != 0 packs W^T (the input-gradient layers of the SDF network). */
uint32_t envidr_packed_weight_floats(int k_order, uint32_t k_in, uint32_t m_out);
uint32_t envidr_packed_rowvec_floats(uint32_t m_out);
int envidr_pack_linear(const float* W_host, uint32_t m_out, uint32_t k_in, int transpose, int k_order,
                       float* dst_host);
/* A whole layer as the fused kernel consumes it: when bias_host != NULL the bias is packed as one extra
 * reduction step placed FIRST (it then travels in the prefetched weight stream; the kernel multiplies
 * it by 1, so the accumulator starts at exactly the bias).  Gradient (transposed) layers have no bias. */
uint32_t envidr_packed_layer_floats(int k_order, uint32_t k_in, uint32_t m_out, int with_bias);
int envidr_pack_layer(const float* W_host, const float* bias_host, uint32_t m_out, uint32_t k_in, int transpose,
                      int k_order, float* dst_host);
/* The SDF network 32 -> 64 -> 64 -> 15 (weights [out, in] row-major + biases) for the 16-column geometry kernel: forward layers,
 * the two transposed layers of the input gradient and row 0 of W3, rows / reduction order permuted as
 * envidr_amd/csrc/geo_eval16.hip.h describes.  dst holds envidr_sdf_geometry_floats() floats. */
uint32_t envidr_sdf_geometry_floats(void);
int envidr_pack_sdf_geometry(const float* W1, const float* b1, const float* W2, const float* b2, const float* W3, const float* b3,
                             float* dst_host);

/* split-precision weight packing (host): W [m_out, k_in] row-major -> (hi, lo) fp16 fragments; k_order as envidr_pack_layer.
 * dst holds envidr_split_layer_halves(k_order, k_in, m_out) uint16. */
uint32_t envidr_split_layer_halves(int k_order, uint32_t k_in, uint32_t m_out);
uint32_t envidr_split_chunk_bytes(void);
uint32_t envidr_split_group(void);
int envidr_pack_layer_split(const float* W_host, uint32_t m_out, uint32_t k_in, int k_order, uint16_t* dst_host);

int envidr_pack_rowvec(const float* v_host, uint32_t m_out, float* dst_host);

/* ---- geometry cache ----------------------------------------------------------
 * For a fixed camera everything up to the compositing weights is independent of the environment: positions, densities,
 * normals, geometry features, roughness.  A geometry_only render with this struct attached appends one record per
 * composited sample (any order; sort by (ray, idx)); envidr_shade_samples + envidr_composite_shaded then re-light the
 * frame for any environment rotation / swapped environment MLP without marching, hash lookups or the SDF network. */
typedef struct envidr_geometry_export {
    uint32_t* counter;     /* device [1], zeroed by the caller; ends at the number of records (may exceed capacity) */
    uint32_t capacity;     /* records the arrays below can hold; records beyond it are counted but not written     */
    uint32_t* ray;         /* [capacity]    ray id                                                                */
    uint32_t* idx;         /* [capacity]    index of the sample within its ray (march order)                       */
    float* w;              /* [capacity]    compositing weight alpha_i * T_i                                        */
    float* normal;         /* [capacity,3]                                                                          */
    float* geo_feat;       /* [capacity,12] unit-normalised                                                         */
    float* roughness;      /* [capacity]                                                                            */
    /* ABI 3: when `slot` is not NULL, normal / geo_feat / roughness / blend of record i live at index slot[i] of their arrays
     * (the geometry pipeline leaves the per-sample data where the evaluation kernel wrote it and hands out indices)        */
    const uint32_t* slot;  /* [capacity] or NULL                                                                            */
    float* blend;          /* raw SDF-network output 14 (learn_indir_blend logit), or NULL                                  */
    /* ABI 6: set by envidr_geometry_pass when it laid its blocks out as 8x8-pixel tiles (desc.image_width), else 0; lets
     * envidr_composite_records walk the rays in the same order, so that a wave's records are again one contiguous run */
    uint32_t image_width;
    /* ABI 7, optional scratch: device uint32 [capacity + capacity / 1024 + 3], or NULL.  With it envidr_shade_records first gathers the
     * records whose compositing weight is NOT exactly zero (shade_list[0] = their number, shade_list[1..] = their indices, ascending) and shades
     * only those; envidr_composite_records skips zero-weight records (w * c = 0 whatever c is).  On a trained scene (beta ~ 1e-3)
     * most samples inside the occupancy shell have alpha = 1 - exp(-sigma * dt) == 0 exactly in fp32; the reference shades them
     * all.  Outputs are unchanged bit for bit.  Ignored by the round-3 form of the split-precision mode (env_split_form 0). */
    uint32_t* shade_list;
} envidr_geometry_export;

/* ---- scene / model description ---------------------------------------------------------------- */
typedef struct envidr_render_desc {
    /* occupancy grid marching (NeRFRenderer state + render kwargs) */
    const uint8_t* density_bitfield; /* device, [cascades * grid_size^3 / 8], Morton order      */
    float bound;                     /* scene half extent; aabb = [-bound, bound]^3              */
    uint32_t cascades;               /* 1 + ceil(log2(bound))                                     */
    uint32_t grid_size;              /* 128                                                       */
    float min_near;                  /* 0.2                                                       */
    uint32_t max_steps;              /* 1024: also the per-ray sample cap                         */
    float dt_gamma;                  /* 0 = constant step                                         */
    float T_thresh;                  /* 1e-4                                                      */
    float density_scale;             /* 1                                                         */
    float bg_color;                  /* scalar background blended with (1 - weights_sum)          */

    /* hash grid (HashEncoder: D = 3, C = 2) */
    const float* hash_table;         /* device, [offsets[L], 2]                                   */
    int32_t hash_offsets[ENVIDR_MAX_LEVELS + 1]; /* HOST copy of the encoder's offsets buffer    */
    uint32_t num_levels;             /* <= 16                                                     */
    uint32_t base_resolution;        /* 16                                                        */
    float log2_per_level_scale;      /* S                                                         */
    int32_t enabled_levels;          /* <= 0: all levels; else features of levels >= this are 0   */

    /* Weights, packed (envidr_pack_linear) and concatenated in CONSUMPTION ORDER into one device blob
     * per pass, each blob zero-padded to a multiple of 4096 floats (one 16 KiB LDS chunk).  The four
     * waves of a workgroup stream a blob through LDS once per pass (envidr_amd/csrc/mlp_mfma.hip.h).
     * Every forward layer is packed WITH its bias (envidr_pack_layer), the two gradient layers without.
     *   sdf_blob  : W1+b (lane order, 32->64) | W2+b (tile, 64->64) | W3+b (k_order 2, 64->15) | W2^T (tile) | W1^T (tile, 64->32)
     *   env_blob  : E1+b (lane, ide_dim->H) | E2+b (tile, H->H) | E3+b (tile, H->H) | E4+b (k_order 2, H->12)
     *   head_blob : D1+b (lane, 24->32) | D2+b (k_order 2, 32->3) | S1+b (lane, 28->64) | S2+b (tile, 64->64) | S3+b (k_order 2, 64->3)
     *   renv_blob : R1+b (lane) | R2+b | R3+b (tile) | R4+b (k_order 2, 64->12);  spec2_blob : S1+b (lane) | S2+b (tile) | S3+b (k_order 2) */
    const float* sdf_blob;
    const float* env_blob;
    const float* head_blob;
    /* SDF network 2*L -> 64 -> 64 -> (1 + 12 + 1 + 1) */
    const float* sdf_w3_row0;        /* row 0 of W3 (d sdf / d h2) as a packed row vector (envidr_pack_rowvec) */
    float beta;                      /* Laplace density beta (already clamped to [beta_min, max]) */
    float roughness_bias;            /* -1                                                        */
    float roughness_act_scale;       /* 0.2                                                       */
    float roughness_scale;           /* 1                                                         */

    /* environment MLP  ide_dim -> H -> H -> H -> 12 (evaluated twice per sample) */
    uint32_t ide_degree;             /* 4 or 5                                                    */
    uint32_t env_hidden;             /* 160 or 256 (multiple of 32)                               */
    float diffuse_kappa_inv;         /* 0.64                                                      */
    float light_intensity_scale;     /* 1                                                         */
    float intensity_scale;           /* 1                                                         */

    /* diffuse head 24 -> 32 -> 3 and specular head 28 -> 64 -> 64 -> 3: in head_blob */

    /* optional environment rotation: w_r and the diffuse normal are multiplied (row vector x
     * matrix) by this row-major 3x3; has_env_rot = 0 skips it (renderer.py:160-161,171-172)      */
    int32_t has_env_rot;
    float env_rot[9];

    /* Network family without an environment MLP (BASELINE configs[1]; network.py:576-584 with use_env_net,
     * use_reflected_dir, diffuse_with_env and wo_viewdir off, encoding_dir = sphere_harmonics):
     *   0     : the environment-MLP family described above;
     *   1..8  : SH "degree" of the view-direction / normal encoders (deg^2 values each).  Then env_blob, ide_degree,
     *           env_hidden and the env rotation are ignored and head_blob is
     *             D1+b (lane, 12->32) | D2+b (tile, 32->3) | S1+b (lane, (2 deg^2 + 13)->64) | S2+b (tile) | S3+b (tile, 64->3)
     *           with specular input [SH(d) | geo_feat | SH(normal) | n.v].  Built: degree 4. */
    uint32_t dir_sh_degree;

    /* Pass selection for indirect-reflection rendering (nerf/renderer.py:437-513 runs the loop three times).
     *   geometry_only != 0 : run_cuda's `geometry_only` branch (cuda_ray.py:303-305): no shading; depth, weights_sum and
     *                        the composited normals (normal_image) are produced, image = (1 - weights_sum) * bg_color.
     *   r_images != NULL   : the main pass with reflected radiance (network.py:612-659,683-690): device [N,4] per ray
     *                        (rgb, visibility) as gathered at cuda_ray.py:291-293.  Samples with roughness <
     *                        indir_roughness_thresh and visibility > 0.9 blend the specular colour with the colour obtained
     *                        from the reflected-radiance features: renv MLP [rgb * vis, sqrt(roughness / roughness_scale /
     *                        0.75)] -> 64 -> 64 -> 64 -> 12, unit-normalised, through the specular head again; blend weight
     *                        0.98 * sigmoid(sdf-network output 14) (learn_indir_blend).
     *   renv_blob  : R1+b (lane, 4->64) | R2+b (tile) | R3+b (tile) | R4+b (tile, 64->12)
     *   spec2_blob : S1+b (lane, 28->64) | S2+b (tile) | S3+b (tile, 64->3)   (the specular head, as in head_blob) */
    int32_t geometry_only;
    const float* r_images;
    const float* renv_blob;
    const float* spec2_blob;
    float indir_roughness_thresh;    /* 0.1 */

    const envidr_geometry_export* geometry_export;   /* HOST pointer or NULL; only with geometry_only != 0 */

    /* Optional scheduling hint, device uint16 [N], caller-owned, read AND written by the call: on entry the number of
     * samples each of these N rays took in an earlier render (zeros if unknown), on return the numbers of this render.
     * The work list is then ordered longest ray first, so the persistent waves run dry together (video frames of one
     * camera: the previous frame's counts are exact).  It changes the order in which rays are processed and nothing
     * else: every output is bit-identical with or without it. */
    uint16_t* ray_cost;

    /* Optional caller-owned work-list scratch (device, >= envidr_render_scratch_bytes(N) bytes, 4-byte aligned), one per
     * render that may be in flight: renders issued on different streams with different scratch (and outputs) overlap --
     * the next frame's waves start on the SIMDs the current frame's finished waves have left.  NULL: the library's own
     * scratch, one render at a time. */
    void* scratch;
    uint64_t scratch_bytes;

    /* ABI 3: the box rays are intersected with (NeRFRenderer.aabb_infer, nerf/renderer.py:76-84; tightened by
     * opt.marching_aabb and restored from checkpoints): {xmin, ymin, zmin, xmax, ymax, zmax}.  has_aabb = 0 means
     * [-bound, bound]^3.  Only the ray / box intersection uses it; sample positions are still clamped to the bound cube. */
    int32_t has_aabb;
    float aabb[6];

    /* ABI 3, geometry pipeline: optional device uint8 [N]; rays whose byte is 0 are not rendered (weights_sum = depth = 0,
     * image = background): the three passes of indirect rendering (renderer.py:439-513) then run over the SAME N rays
     * with masks instead of boolean-mask gathers, host-side counts and scatters between them. */
    const uint8_t* ray_mask;

    /* ABI 4, optional split-precision shading mode (NOT the default, never what bench.py's headline runs): when
     * env_split_blob is set, envidr_shade_samples / envidr_shade_records evaluate the environment MLP on the fp16 matrix
     * cores with every operand carried as a (hi, lo) fp16 pair -- 22-bit significands, fp32 accumulation
     * (envidr_amd/csrc/mlp_split.hip.h) -- and only the heads in fp32.
     *   env_split_blob  device: the four layers packed by envidr_pack_layer_split in consumption order, zero-padded to a
     *                   multiple of envidr_split_chunk_bytes()
     *   env_split_bias  device float: the four biases as envidr_pack_rowvec tiles, concatenated
     *   env_features    device float [capacity, 24] scratch the mode writes env(normal) | env(reflection) into */
    const void* env_split_blob;
    const float* env_split_bias;
    float* env_features;

    /* ABI 5, optional: the SDF network packed for the 16-column geometry kernel (envidr_pack_sdf_geometry; device,
     * envidr_sdf_geometry_floats() floats).  With it the geometry entry points run k_geo_eval16 (16 samples per wave, three
     * waves per SIMD); without it k_geo_eval32 on sdf_blob.  Same results up to fp32 summation order. */
    const float* sdf_geo_blob;

    /* ABI 6, optional, geometry pipeline only: the rays of this call are the pixels of a row-major image `image_width` pixels
     * wide (N a multiple of it, width and height multiples of 8).  A layout hint: the pipeline then forms its blocks of 64 rays
     * from 8x8-pixel tiles instead of 64 consecutive pixels of a row, so that the samples a wave evaluates together are
     * neighbours in both image directions (their hash gathers share more lines: -8 % geometry time at 800x800).  Every output
     * is still indexed by the ray's position in the call's list and has the same bits; 0 (or a size that does not qualify) =
     * list order. */
    uint32_t image_width;

    /* ABI 10: which kernel evaluates the split-precision mode, i.e. what env_split_blob holds.
     *   0  the round-3 form (csrc/shade_split.hip): a wave's two groups of 32 items one after the other; blob = the four layers packed by
     *      envidr_pack_layer_split in consumption order
     *   1  the fused-pair form (csrc/shade_split2.hip): layers fused in pairs, eight waves of 256 registers on one weight stream; blob packed by
     *      envidr_pack_env_split2 (envidr_env_split2_halves() halves).  Same results bit for bit.  Honours geometry_export.shade_list. */
    uint32_t env_split_form;
} envidr_render_desc;

/* ---- per-call outputs (device pointers; any optional pointer may be NULL) --------------------- */
typedef struct envidr_render_out {
    float* image;           /* [N,3]  composited rgb + (1 - weights_sum) * bg_color                */
    float* depth;           /* [N]                                                                 */
    float* weights_sum;     /* [N]                                                                 */
    float* normal_image;    /* [N,3]  optional: normalize(sum w n) (eps 1e-10), cuda_ray.py:357    */
    float* diffuse_image;   /* [N,3]  optional                                                     */
    float* specular_image;  /* [N,3]  optional                                                     */
    float* roughness_image; /* [N]    optional: sum w roughness                                    */
    uint64_t* stats;        /* [12]   optional: {samples shaded, wave rounds, rays, reserved, 8 x section cycles
                             *         (only when the library is built with -DENVIDR_SECTION_TIMERS)}; caller zeroes */
} envidr_render_out;

uint64_t envidr_render_scratch_bytes(uint32_t N);

/* Render N rays (rays_o, rays_d: device [N,3], unit directions).  `ray_counter` is a device uint32
 * the kernel uses as its work queue head; the call zeroes it on `stream` before launching.
 * Sample positions, occupancy decisions and step sizes equal the reference loop run with one sample
 * per ray per iteration; colours agree with the reference's fp32 PyTorch path to fp32 rounding
 * (DESIGN.md "parity"). */
int envidr_render_rays(const envidr_render_desc* desc, const float* rays_o, const float* rays_d, uint32_t N,
                       const envidr_render_out* out, uint32_t* ray_counter, envidr_stream_t stream);

/* Shade M samples whose geometry is already known: the per-sample shading of the loop
 *     nerf/renderer.py:147-180   get_color_mlp_extra_params (reflection, env rotation, IDE x2, n.v)
 *     nerf/network.py:524-698    forward_color               (env MLP x2, diffuse + specular heads)
 * without marching, hash grid and SDF network.  This is the whole of demo.ipynb cell 17 after the
 * ray / sphere intersection (surface rendering, BASELINE configs[0]) and the per-frame part of re-lighting
 * or rotating the environment around cached geometry.  Uses desc->env_blob, head_blob, ide_degree, env_hidden,
 * diffuse_kappa_inv, light_intensity_scale and the env rotation; every other field is ignored.
 *   normals, dirs : device [M,3], unit;  dirs = view direction (camera -> sample)
 *   geo_feat      : device [M,12] (stride 12) or ONE shared [12] (stride 0), unit-normalised
 *   roughness     : device [M] (stride 1) or ONE shared value (stride 0): kappa_inv of the reflected-direction IDE
 *   c_diffuse, c_specular : device [M,3] out, sigmoid colours (the caller adds them and applies intensity_scale) */
int envidr_shade_samples(const envidr_render_desc* desc, const float* normals, const float* dirs, const float* geo_feat,
                         uint32_t geo_feat_stride, const float* roughness, uint32_t roughness_stride, uint32_t M,
                         float* c_diffuse, float* c_specular, envidr_stream_t stream);

/* The environment MLP alone, for callers that keep the reference's operator loop (IDE encoder, MLPs and compositor as separate
 * calls): y = W4 relu(W3 relu(W2 relu(W1 x + b1) + b2) + b3) + b4 -- the Linear / ReLU chain of nerf/network.py:533-536 and
 * 592-595 (env_net, built by get_env_net, network.py:279-297) -- on the fp32 matrix cores, 64 rows per wave round, the pass the shading kernels run.
 *   env_blob : device, the four layers packed with envidr_pack_layer in the orders envidr_render_desc.env_blob uses
 *   x        : device [M, in_dim] IDE codes (in_dim = 72 for IDE degree 5, 38 for degree 4);  y : device [M, 12], raw (not normalised)
 * Built for (in_dim, hidden) = (72,256) (38,160) (72,128) (38,128); anything else is ENVIDR_EINVAL. */
int envidr_env_mlp_forward(const float* env_blob, uint32_t in_dim, uint32_t hidden, const float* x, uint32_t M, float* y,
                           envidr_stream_t stream);

/* Two-phase frames (geometry pass -> shading pass): the records of a geometry_only render are shaded where they lie, in the
 * order they were appended (the count is read on the device: the host does not wait for the geometry pass), and
 * composited through a permutation built from the per-ray sample counts.
 *   envidr_shade_records    : c_diffuse / c_specular [capacity,3] for records 0 .. min(*counter, capacity) - 1;
 *                             view direction of record i = rays_d[ray[i]]
 *   envidr_composite_records: offsets = exclusive prefix sum of the per-ray counts (device uint32 [N+1]; the counts are what
 *                             desc.ray_cost holds after the geometry pass); perm = device scratch uint32 [capacity];
 *                             then as envidr_composite_shaded.  Same bits as envidr_render_rays on the same rays. */
int envidr_shade_records(const envidr_render_desc* desc, const envidr_geometry_export* records, const float* rays_d,
                         float* c_diffuse, float* c_specular, envidr_stream_t stream);
int envidr_composite_records(const envidr_geometry_export* records, const uint32_t* offsets, uint32_t* perm, const float* c_diffuse,
                             const float* c_specular, const float* weights_sum, uint32_t N, float intensity_scale, float bg_color,
                             float* image, float* diffuse_image, float* specular_image, envidr_stream_t stream);

/* ---- geometry pipeline (envidr_amd/csrc/geometry_pass.hip) -------------------------------------------------------
 * The per-SAMPLE half of the geometry pass as its own operator: for M sample positions, the 16-level hash grid with
 * analytic Jacobian, the SDF network forward and input gradient, and the per-sample terms derived from them --
 *     nerf/network.py:381-522   forward_geometry / forward_sigma   (hash grid -> SDF MLP -> Laplace density)
 *     nerf/renderer.py:182-198  compute_normal                      (analytic instead of autograd)
 * One lane per sample, two waves per SIMD, SDF weights resident in LDS.  Same per-sample arithmetic (same bits) as
 * envidr_render_rays.  Uses desc->hash_*, sdf_blob, sdf_w3_row0, beta, density_scale, roughness_*, bound.
 *   xyz   : device [M,3] positions;  dt : device [M] step sizes (only for `alpha`; may be NULL when alpha is NULL)
 *   range : optional device uint32 {begin, count}: evaluate samples [begin, begin + count) of arrays holding M slots
 *           (count read on the device: the host never waits for the kernel that produced the samples); NULL = [0, M)
 * Any output pointer may be NULL. */
typedef struct envidr_geometry_samples_out {
    float* alpha;      /* [M]    1 - exp(-sigma dt)                                   */
    float* sigma;      /* [M]    Laplace density * density_scale                        */
    float* normal;     /* [M,3]  unit (eps 1e-10)                                       */
    float* geo_feat;   /* [M,12] unit (eps 1e-12)                                       */
    float* roughness;  /* [M]                                                           */
    float* blend;      /* [M]    raw SDF-network output 14 (learn_indir_blend logit)    */
} envidr_geometry_samples_out;

int envidr_geometry_eval(const envidr_render_desc* desc, const float* xyz, const float* dt, uint32_t M,
                         const uint32_t* range_dev, const envidr_geometry_samples_out* out, envidr_stream_t stream);

/* ABI 6, TEST HOOK: what the hash section and the SDF network of the geometry kernel (k_geo_eval32, the kernel every frame of
 * the pipeline runs) compute for positions xyz [M,3], before any per-sample term is derived from it -- so that the benchmarked
 * gather path can be compared with the reference's index arithmetic (hashencoder/src/hashencoder.cu:36-69,103-254) as integers
 * and per value, not only through rendered images.  Any output may be NULL:
 *   features     [M,32]    hash features as they enter the network (level-major, 2 channels; zeros outside the unit cube)
 *   corner_rows  [M,16,8]  row within its level of corner bx | by << 1 | bz << 2 (the reference's `(index % hashmap_size)`);
 *                          for positions outside the cube the rows of the clamped stand-in position (their features are masked)
 *   raw_outputs  [M,16]    the last SDF layer's outputs 0..15 (0 = sdf, 1..12 = geo_feat before normalisation, 13 = roughness
 *                          logit, 14 = blend logit, 15 = padding)
 *   sdf_gradient [M,3]     d sdf / d xyz before normalisation */
int envidr_geometry_probe(const envidr_render_desc* desc, const float* xyz, uint32_t M, float* features, uint32_t* corner_rows,
                          float* raw_outputs, float* sdf_gradient, envidr_stream_t stream);

/* The geometry half of a frame as a device-driven pipeline (two-phase frames; envidr_amd/csrc/geometry_pass.hip):
 *     nerf/render_func/cuda_ray.py:277-346  the march -> evaluate -> composite -> compact loop, geometry part
 * Rays march in chunks (16, 32, ... samples); each round one per-ray kernel composites the previous chunk (appending
 * one record per composited sample) and marches the next, and one per-sample kernel (envidr_geometry_eval) evaluates
 * the new samples.  All counts stay on the device; the call only enqueues.  Sample positions, step sizes and occupancy
 * decisions are those of envidr_render_rays (the reference loop with one sample per iteration).
 *   out      : depth, weights_sum (required), normal_image, roughness_image; out->stats (optional, uint64[3]):
 *              {samples evaluated, records, overflow code}.  desc->ray_cost (uint16 [N], optional) receives the number of
 *              composited samples per ray (what envidr_composite_records' offsets are the prefix sum of).
 *   records  : caller-owned counter / ray / idx / w / slot arrays of `capacity` entries; on return the struct's normal /
 *              geo_feat / roughness / blend pointers address the per-sample arrays inside the workspace.  If the frame did
 *              not fit (samples or records), *counter is 0xffffffff: envidr_shade_records / envidr_composite_records then
 *              do nothing and the caller redoes the frame with larger buffers.
 *   workspace: device, 256-byte aligned, >= envidr_geometry_workspace_bytes(N, sample_capacity) bytes; may be reused by
 *              the next frame once the records have been shaded. */
uint64_t envidr_geometry_workspace_bytes(uint32_t N, uint32_t sample_capacity);
int envidr_geometry_pass(const envidr_render_desc* desc, const float* rays_o, const float* rays_d, uint32_t N,
                         const envidr_render_out* out, envidr_geometry_export* records, void* workspace,
                         uint64_t workspace_bytes, uint32_t sample_capacity, envidr_stream_t stream);

/* Composite shaded colours over cached geometry: ray r owns records offsets[r] .. offsets[r+1]-1 (sorted by (ray, idx)).
 *   image[r] = sum_i w_i (c_diffuse_i + c_specular_i) intensity_scale + (1 - weights_sum[r]) bg_color, and the optional
 *   diffuse / specular images = sum_i w_i c_i: the blend of the render loop (cuda_ray.py:318-340) with known weights,
 *   same accumulation order. */
int envidr_composite_shaded(const uint32_t* offsets, const float* w, const float* c_diffuse, const float* c_specular,
                            const float* weights_sum, uint32_t N, float intensity_scale, float bg_color, float* image,
                            float* diffuse_image, float* specular_image, envidr_stream_t stream);

/* ---- ABI 8: env-sphere mode (nerf/render_func/sph_ray.py:34-221, `run_sph`; selected by opt.env_sph_mode, renderer.py:376) ----------
 * The object is a sphere of known radius: a ray's hit is analytic (get_sphere_intersections, sph_ray.py:18-32), every hit ray gets S samples spaced step_size around its hit, and the samples are composited with the torch formulation of
 * volume rendering (alphas + cumulative product), not the marcher's compositing kernel.  The SDF network / shading of the samples are
 * envidr_geometry_eval and envidr_shade_samples.  Per-sample arrays are SAMPLE-major: [S, M, ...].
 *
 * envidr_sphere_intersections (sph_ray.py:18-32): near = -d.o - sqrt(max(D, 0)), far = -d.o + sqrt(max(D, 0)), D = (d.o)^2 - (|o|^2 - r^2),
 *   mask = D >= -1e-4; rays_o, rays_d [N,3] (unit directions); nears, fars [N] float, mask [N] uint8.
 *
 * envidr_shell_samples (sph_ray.py:69-79): z = z_offsets[s] + near (+ (noise - 0.5) step_size), xyz = o + d z.
 *   hit_rays [M] int32: ids of the rays that hit, ascending;  nears [N];  z_offsets [S] = linspace(-r, r, S), r = step_size (S-1) / 2;
 *   noise [M, S] in [0, 1) or NULL (perturb);  out: xyz [S,M,3], dirs [S,M,3] (the ray direction, repeated), z_vals [S,M]. */
int envidr_sphere_intersections(const float* rays_o, const float* rays_d, uint32_t N, float radius, float* nears, float* fars, uint8_t* mask,
                                envidr_stream_t stream);
int envidr_shell_samples(const float* rays_o, const float* rays_d, const int32_t* hit_rays, const float* nears, const float* z_offsets,
                         const float* noise, float step_size, uint32_t M, uint32_t S, float* xyz, float* dirs, float* z_vals,
                         envidr_stream_t stream);

/* envidr_composite_shell (sph_ray.py:102-151): deltas = z[s+1] - z[s] (last: step_size), alpha = 1 - exp(-delta sigma),
 * w = alpha prod_{s' < s} (1 - alpha_s' + 1e-15); per ray n with hit_slot[n] = m >= 0:
 *   weights_sum = sum w;  depth = sum w clamp((z - near) / (*far_max - near), 0, 1);
 *   image = sum w (c_diffuse + c_specular) intensity_scale + (1 - weights_sum) bg[n];  diffuse / specular images likewise with their colour;
 *   normal_image = normalize(sum w normal) (eps 1e-12);  roughness_image = sum w roughness;
 * rays with hit_slot[n] < 0 get bg (image, diffuse, specular) and zeros (depth, weights_sum, normal, roughness).
 *   sigma, z_vals, roughness [S,M]; c_diffuse, c_specular, normals [S,M,3]; hit_slot, nears [N]; far_max: device scalar (fars.max() over
 *   ALL rays, sph_ray.py:112); bg [N,3].  normals / roughness and the four optional images may be NULL. */
int envidr_composite_shell(const float* sigma, const float* z_vals, const float* c_diffuse, const float* c_specular, const float* normals,
                           const float* roughness, const int32_t* hit_slot, const float* nears, const float* far_max, const float* bg,
                           uint32_t N, uint32_t M, uint32_t S, float step_size, float intensity_scale, float* image, float* depth,
                           float* weights_sum, float* normal_image, float* diffuse_image, float* specular_image, float* roughness_image,
                           envidr_stream_t stream);

/* ---- ABI 8: weight gradient of a dense layer over a large batch (training branch, reference cuda_ray.py:64-237: what torch autograd asks of
 * nn.Linear there; the reference leaves it to cuBLAS) -------------------------------------------------------------------------------------
 *   dW[o][i] (+)= sum_m gy[m][o] x[m][i]      db[o] (+)= sum_m gy[m][o]        x [M, K_in], gy [M, N_out] row-major, 16-byte aligned
 * The reduction over the samples is split across the chip (fp32 MFMA, per-chunk partial results in `workspace`, summed in a fixed order:
 * deterministic).  db may be NULL; accumulate != 0 adds to dW / db.  workspace: device, 16-byte aligned,
 * >= envidr_linear_weight_grad_workspace_bytes(M, K_in, N_out). */
uint64_t envidr_linear_weight_grad_workspace_bytes(uint32_t M, uint32_t K_in, uint32_t N_out);
int envidr_linear_weight_grad(const float* x, const float* gy, uint32_t M, uint32_t K_in, uint32_t N_out, float* dW, float* db, int accumulate,
                              void* workspace, uint64_t workspace_bytes, envidr_stream_t stream);

/* ---- ABI 9: a dense layer over a large batch of rows (the forward and input-gradient products of the same training branch; the reference:
 * cuBLAS plus one elementwise kernel per bias / ReLU / ReLU gradient) -----------------------------------------------------------------------
 *   y[m][o] = epilogue( sum_i x[m][i] W(o, i) ),   W(o, i) = W[o * w_stride_out + i * w_stride_in]          m < M, i < K, o < N
 * so W [N][K] row-major (strides K, 1) gives y = x W^T, and the same matrix read with strides (1, N') gives the input gradient gy W without a
 * transposed copy.  Epilogues:  PLAIN none;  BIAS + bias[o];  BIAS_RELU max(. + bias[o], 0);  RELU_MASK . * (act[m][o] > 0) -- the ReLU
 * gradient for the activation `act` this gradient flows back through.  fp32 on the matrix cores.  x rows 16-byte aligned (ldx % 4 == 0),
 * K a multiple of 4; ldx / ldact / ldy are row pitches in floats.  bias / act may be NULL where the epilogue does not read them. */
enum { ENVIDR_ROWS_PLAIN = 0, ENVIDR_ROWS_BIAS = 1, ENVIDR_ROWS_BIAS_RELU = 2, ENVIDR_ROWS_RELU_MASK = 3 };
int envidr_linear_rows(const float* x, uint32_t ldx, uint32_t M, uint32_t K, const float* W, int64_t w_stride_out, int64_t w_stride_in, uint32_t N,
                       const float* bias, const float* act, uint32_t ldact, int epilogue, float* y, uint32_t ldy, envidr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ENVIDR_RENDER_H */
