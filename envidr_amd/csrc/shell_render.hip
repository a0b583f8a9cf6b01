// Env-sphere mode (reference nerf/render_func/sph_ray.py:34-221, `run_sph`): the object is a sphere whose ray hits are analytic;
// every hit ray gets S samples (12) spaced `step_size` (0.002) around its hit, the SDF network (with the material parameters
// concatenated to the hash features) gives density, normal, features and roughness there, the samples are shaded like any
// other and composited with the TORCH formulation of volume rendering (alphas, a cumulative product of 1 - alpha + 1e-15) --
// not the marcher's compositing kernel.  Two operators bracket the shared geometry / shading kernels:
//     envidr_shell_samples     sph_ray.py:69-79     z_vals, xyzs, dirs of the S x M samples
//     envidr_composite_shell   sph_ray.py:102-151   deltas, alphas, weights, images, depth, un-masking into the N rays
// Per-sample arrays are SAMPLE-major ([S, M, ...]: sample s of all hit rays, then sample s + 1): one lane per hit ray reads and
// writes contiguous runs, and the 64 positions a wave of the geometry kernel evaluates together lie at the same depth of
// neighbouring rays, i.e. in neighbouring cells of the hash grid.
#include "common.hip.h"
#include "../../include/envidr_render.h"

namespace envidr {
namespace {

// get_sphere_intersections (sph_ray.py:18-32): near / far parameters of |x| = r along o + t d (d unit), the discriminant clamped at 0
// under the root; a ray counts as a hit from a discriminant of -1e-4 (grazing rays).  One lane per ray.  The reference forms d.o with
// torch.bmm -- 640 000 one-by-three times three-by-one products cost 8.8 ms per 800 x 800 frame on this GPU.  The expressions below are
// the reference's, in fp32, in its order, product by product (r^2 arrives as the float the reference's Python scalar r ** 2 becomes): the
// hit set of the reference's CPU run is reproduced on the fixtures (tests/test_sph_gpu.py), which an fp64 evaluation does NOT do -- the
// discriminant (d.o)^2 - (|o|^2 - r^2) cancels ~95 % of its terms for a camera 4 radii away, its fp32 rounding (1e-6) decides the -1e-4
// test for a handful of silhouette rays per frame, and those rays are pixels.  Both forms of run_sph take their hits from this kernel.
__global__ void __launch_bounds__(kBlock) k_sphere_intersections(const float* __restrict__ rays_o, const float* __restrict__ rays_d, uint32_t N,
                                                                 float radius2, float* __restrict__ nears, float* __restrict__ fars,
                                                                 uint8_t* __restrict__ mask) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    const float dot = dx * ox + dy * oy + dz * oz;                                   // ray_cam_dot
    const float len = sqrtf(ox * ox + oy * oy + oz * oz);                            // rays_o.norm(2, 1)
    const float nabla = dot * dot - (len * len - radius2);
    const float root = sqrtf(fmaxf(nabla, 0.0f));
    nears[n] = -dot - root;
    fars[n] = -dot + root;
    mask[n] = nabla >= -1e-4f ? 1 : 0;
}

__global__ void __launch_bounds__(kBlock) k_shell_samples(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                          const int32_t* __restrict__ hit_rays, const float* __restrict__ nears,
                                                          const float* __restrict__ z_offsets, const float* __restrict__ noise,
                                                          float step_size, uint32_t M, uint32_t S, float* __restrict__ xyz,
                                                          float* __restrict__ dirs, float* __restrict__ z_vals) {
    const uint32_t m = blockIdx.x * kBlock + threadIdx.x;
    if (m >= M) return;
    const uint32_t ray = (uint32_t)hit_rays[m];
    const float ox = rays_o[ray * 3], oy = rays_o[ray * 3 + 1], oz = rays_o[ray * 3 + 2];
    const float dx = rays_d[ray * 3], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
    const float near = nears[ray];
    for (uint32_t s = 0; s < S; ++s) {
        float z = z_offsets[s] + near;                                               // sph_ray.py:72-73
        if (noise) z = z + (noise[(size_t)m * S + s] - 0.5f) * step_size;           // :75-76 (perturb)
        const size_t i = (size_t)s * M + m;
        z_vals[i] = z;
        xyz[i * 3] = ox + dx * z;                                                    // :79 (product, then sum: no contraction)
        xyz[i * 3 + 1] = oy + dy * z;
        xyz[i * 3 + 2] = oz + dz * z;
        dirs[i * 3] = dx; dirs[i * 3 + 1] = dy; dirs[i * 3 + 2] = dz;
    }
}

struct ShellCompositeArgs {
    const float* sigma;        // [S, M]   density (density_scale applied)
    const float* z_vals;       // [S, M]
    const float* c_diffuse;    // [S, M, 3]
    const float* c_specular;   // [S, M, 3]
    const float* normals;      // [S, M, 3] or null
    const float* roughness;    // [S, M] or null
    const int32_t* hit_slot;   // [N]  index m of the ray among the hit rays, -1 for a ray that misses the sphere
    const float* nears;        // [N]
    const float* far_max;      // device scalar: max over ALL rays of `far` (sph_ray.py:112)
    const float* bg;           // [N, 3]
    uint32_t N, M, S;
    float step_size, intensity_scale;
    float *image, *depth, *weights_sum, *normal_image, *diffuse_image, *specular_image, *roughness_image;
};

__global__ void __launch_bounds__(kBlock) k_composite_shell(const ShellCompositeArgs a) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= a.N) return;
    const float b0 = a.bg[n * 3], b1 = a.bg[n * 3 + 1], b2 = a.bg[n * 3 + 2];
    const int32_t m = a.hit_slot[n];
    if (m < 0) {                                                   // un-masking (sph_ray.py:114,120,125-150): background / zeros
        a.image[n * 3] = b0; a.image[n * 3 + 1] = b1; a.image[n * 3 + 2] = b2;
        a.depth[n] = 0; a.weights_sum[n] = 0;
        if (a.normal_image) { a.normal_image[n * 3] = 0; a.normal_image[n * 3 + 1] = 0; a.normal_image[n * 3 + 2] = 0; }
        if (a.diffuse_image) { a.diffuse_image[n * 3] = b0; a.diffuse_image[n * 3 + 1] = b1; a.diffuse_image[n * 3 + 2] = b2; }
        if (a.specular_image) { a.specular_image[n * 3] = b0; a.specular_image[n * 3 + 1] = b1; a.specular_image[n * 3 + 2] = b2; }
        if (a.roughness_image) a.roughness_image[n] = 0;
        return;
    }
    const float near = a.nears[n];
    const float inv_range_den = *a.far_max - near;                 // (z - near) / (fars.max() - near), clamped to [0, 1]
    float T = 1.0f;                                                // cumprod([1, 1 - alpha + 1e-15 ...])[:-1]
    float ws = 0, dep = 0, rgb[3] = {0, 0, 0}, dif[3] = {0, 0, 0}, spc[3] = {0, 0, 0}, nrm[3] = {0, 0, 0}, rgh = 0;
    float z = a.z_vals[m];
    for (uint32_t s = 0; s < a.S; ++s) {
        const size_t i = (size_t)s * a.M + (uint32_t)m;
        const float z_next = s + 1 < a.S ? a.z_vals[i + a.M] : 0.0f;
        const float delta = s + 1 < a.S ? z_next - z : a.step_size;                         // sph_ray.py:103-104
        const float alpha = 1.0f - expf(-delta * a.sigma[i]);                               // :105
        const float w = alpha * T;                                                          // :107
        T = T * (1.0f - alpha + 1e-15f);                                                    // :106
        ws += w;
        float o = (z - near) / inv_range_den;
        o = fminf(fmaxf(o, 0.0f), 1.0f);
        dep += w * o;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float cd = a.c_diffuse[i * 3 + c], cs = a.c_specular[i * 3 + c];
            rgb[c] += w * ((cd + cs) * a.intensity_scale);                                  // forward_color's return value (network.py:698)
            dif[c] += w * cd;
            spc[c] += w * cs;
            if (a.normals) nrm[c] += w * a.normals[i * 3 + c];
        }
        if (a.roughness) rgh += w * a.roughness[i];
        z = z_next;
    }
    const float rest = 1.0f - ws;
    a.image[n * 3] = rgb[0] + rest * b0; a.image[n * 3 + 1] = rgb[1] + rest * b1; a.image[n * 3 + 2] = rgb[2] + rest * b2;   // :118-119
    a.depth[n] = dep;
    a.weights_sum[n] = ws;
    if (a.normal_image) {                                                                   // :122-124, F.normalize default eps 1e-12
        const float len = fmaxf(sqrtf(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]), 1e-12f);
        a.normal_image[n * 3] = nrm[0] / len; a.normal_image[n * 3 + 1] = nrm[1] / len; a.normal_image[n * 3 + 2] = nrm[2] / len;
    }
    if (a.diffuse_image) { a.diffuse_image[n * 3] = dif[0] + rest * b0; a.diffuse_image[n * 3 + 1] = dif[1] + rest * b1; a.diffuse_image[n * 3 + 2] = dif[2] + rest * b2; }
    if (a.specular_image) { a.specular_image[n * 3] = spc[0] + rest * b0; a.specular_image[n * 3 + 1] = spc[1] + rest * b1; a.specular_image[n * 3 + 2] = spc[2] + rest * b2; }
    if (a.roughness_image) a.roughness_image[n] = rgh;
}

}  // namespace
}  // namespace envidr

using namespace envidr;

extern "C" {

int envidr_sphere_intersections(const float* rays_o, const float* rays_d, uint32_t N, float radius, float* nears, float* fars, uint8_t* mask,
                                envidr_stream_t stream) {
    if (N == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(rays_o && rays_d && nears && fars && mask, "sphere_intersections: null pointer");
    const float radius2 = (float)((double)radius * (double)radius);           // torch: the Python scalar r ** 2 (a double) cast to the tensor's float
    hipLaunchKernelGGL(k_sphere_intersections, dim3(ceil_div(N, kBlock)), dim3(kBlock), 0, as_stream(stream), rays_o, rays_d, N, radius2, nears, fars, mask);
    return check_launch("k_sphere_intersections");
}

int envidr_shell_samples(const float* rays_o, const float* rays_d, const int32_t* hit_rays, const float* nears, const float* z_offsets,
                         const float* noise, float step_size, uint32_t M, uint32_t S, float* xyz, float* dirs, float* z_vals,
                         envidr_stream_t stream) {
    if (M == 0 || S == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(rays_o && rays_d && hit_rays && nears && z_offsets && xyz && dirs && z_vals, "shell_samples: null pointer");
    hipLaunchKernelGGL(k_shell_samples, dim3(ceil_div(M, kBlock)), dim3(kBlock), 0, as_stream(stream), rays_o, rays_d, hit_rays, nears,
                       z_offsets, noise, step_size, M, S, xyz, dirs, z_vals);
    return check_launch("k_shell_samples");
}

int envidr_composite_shell(const float* sigma, const float* z_vals, const float* c_diffuse, const float* c_specular, const float* normals,
                           const float* roughness, const int32_t* hit_slot, const float* nears, const float* far_max, const float* bg,
                           uint32_t N, uint32_t M, uint32_t S, float step_size, float intensity_scale, float* image, float* depth,
                           float* weights_sum, float* normal_image, float* diffuse_image, float* specular_image, float* roughness_image,
                           envidr_stream_t stream) {
    if (N == 0) return ENVIDR_OK;
    ENVIDR_REQUIRE(hit_slot && nears && far_max && bg && image && depth && weights_sum, "composite_shell: null pointer");
    ENVIDR_REQUIRE(M == 0 || (sigma && z_vals && c_diffuse && c_specular), "composite_shell: null sample array");
    ENVIDR_REQUIRE(S >= 1, "composite_shell: S must be >= 1");
    ENVIDR_REQUIRE(!normal_image || normals, "composite_shell: normal_image needs the per-sample normals");
    ENVIDR_REQUIRE(!roughness_image || roughness, "composite_shell: roughness_image needs the per-sample roughness");
    ShellCompositeArgs a{sigma, z_vals, c_diffuse, c_specular, normals, roughness, hit_slot, nears, far_max, bg, N, M, S, step_size,
                         intensity_scale, image, depth, weights_sum, normal_image, diffuse_image, specular_image, roughness_image};
    hipLaunchKernelGGL(k_composite_shell, dim3(ceil_div(N, kBlock)), dim3(kBlock), 0, as_stream(stream), a);
    return check_launch("k_composite_shell");
}

}  // extern "C"
