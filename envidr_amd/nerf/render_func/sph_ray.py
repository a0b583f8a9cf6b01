"""Surface rendering of the environment-lit sphere: BASELINE configs[0] (the reference's `demo.ipynb`, "Rendering Steps"
cell) and the ray / sphere intersection its env-sphere mode is built on (nerf/render_func/sph_ray.py:18-32).

Two forms of it live here:

* `render_surface` -- the notebook's form: one surface sample per hit ray, shaded by IDE x2 + environment MLP x2 + diffuse /
  specular heads, which is exactly `envidr_shade_samples` (`FusedShader.shade`, csrc/fused_render.hip k_shade_samples).
* `run_sph` (round 5) -- the reference's own env-sphere render function (sph_ray.py:34-221, selected by `opt.env_sph_mode`,
  renderer.py:376-377; configs/neural_renderer.ini is how the shipped rendering MLPs were trained): 12 samples spaced 0.002
  around every analytic hit, the SDF network with the call's material parameters concatenated to the hash features
  (network.py:165-175, 369-379), shading, and the torch formulation of volume rendering.  On the GPU, in eval mode, for the
  network shapes the fused kernels are built for, it runs as four launches -- `envidr_shell_samples` ->
  `envidr_geometry_eval` -> `envidr_shade_samples` -> `envidr_composite_shell` (csrc/shell_render.hip); otherwise as the
  reference's chain of operators (`forward_sigma` / `get_color_mlp_extra_params` / `forward_color` + torch compositing).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def get_sphere_intersections(rays_o: torch.Tensor, rays_d: torch.Tensor, r: float = 1.0):
    """near / far ray parameters of the sphere |x| = r and the mask of rays that reach it; rays_o, rays_d [N,3] (unit
    directions) -> near [N,1], far [N,1], mask [N].  Reference sph_ray.py:18-32: discriminant clamped at 0 under the root,
    a ray counts as a hit from a discriminant of -1e-4 (grazing rays)."""
    if rays_o.is_cuda:
        # on the GPU: envidr_sphere_intersections (csrc/shell_render.hip: the same fp32 expressions in the same order, one lane per ray;
        # torch.bmm of N 1x3 by 3x1 products costs 8.8 ms per 800 x 800 frame)
        from ... import _lib
        from ...fused import _bind_render
        lib = _lib.load()
        _bind_render(lib)
        o, d = rays_o.reshape(-1, 3).float().contiguous(), rays_d.reshape(-1, 3).float().contiguous()
        N = o.shape[0]
        near, far = torch.empty(N, 1, device=o.device), torch.empty(N, 1, device=o.device)
        mask = torch.empty(N, dtype=torch.uint8, device=o.device)
        rc = lib.envidr_sphere_intersections(o.data_ptr(), d.data_ptr(), N, float(r), near.data_ptr(), far.data_ptr(), mask.data_ptr(),
                                             torch.cuda.current_stream(o.device).cuda_stream)
        if rc:
            raise _lib.EnvidrError(f"envidr_sphere_intersections failed ({rc}): {lib.envidr_last_error().decode()}")
        return near, far, mask.bool()
    ray_cam_dot = torch.bmm(rays_d.view(-1, 1, 3), rays_o.view(-1, 3, 1)).squeeze(-1)
    nabla = ray_cam_dot ** 2 - (rays_o.norm(2, 1, keepdim=True) ** 2 - r ** 2)
    nabla_sqrt = torch.sqrt(nabla.clamp_min(0.0))
    return -ray_cam_dot - nabla_sqrt, -ray_cam_dot + nabla_sqrt, (nabla >= -1e-4)[..., 0]


def material_features(sdf_layers, xyz_encoding: torch.Tensor, roughness: float, metallic: float, base_color, feat_dim: int = 12,
                      roughness_bias: float = -1.0, roughness_act_scale: float = 1.0):
    """the notebook's material network on its single constant position feature: [xyz_encoding | roughness, metallic, rgb] ->
    sdf_net -> (unit geo_feat [feat_dim], kappa_inv scalar).  One 37-vector through three tiny layers: host-side torch."""
    dev = xyz_encoding.device
    h = torch.cat([xyz_encoding.reshape(-1), torch.tensor([roughness, metallic, *base_color], dtype=torch.float32, device=dev)])[None]
    for i, (W, b) in enumerate(sdf_layers):
        h = F.linear(h, torch.as_tensor(W, device=dev), torch.as_tensor(b, device=dev))
        if i < len(sdf_layers) - 1:
            h = F.relu(h)
    geo_feat = F.normalize(h[..., 1:1 + feat_dim], dim=-1)[0]
    kappa_inv = roughness_act_scale * F.softplus(h[..., -1] + roughness_bias)[0]
    return geo_feat, kappa_inv


def render_surface(shader, rays_o: torch.Tensor, rays_d: torch.Tensor, geo_feat: torch.Tensor, kappa_inv, radius: float = 1.0,
                   bg_color: float = 1.0, env_rot_radian: float | None = None) -> dict:
    """One surface sample per ray that hits the sphere, shaded on the GPU (`shader`: envidr_amd.fused.FusedShader with the
    environment MLP + heads resident), composed over `bg_color` like the notebook does.  rays [N,3] on the GPU.
    Returns image / diffuse_image / specular_image [N,3], mask [N], depth [N] (0 where the sphere is missed)."""
    rays_o = rays_o.contiguous().view(-1, 3).float()
    rays_d = rays_d.contiguous().view(-1, 3).float()
    near, _, mask = get_sphere_intersections(rays_o, rays_d, radius)
    dirs = rays_d[mask]
    xyzs = rays_o[mask] + dirs * near[mask]
    normals = xyzs if radius == 1.0 else xyzs / radius            # the hit point IS the normal on the unit sphere (notebook)
    N = rays_o.shape[0]
    bg = torch.zeros(N, 3, device=rays_o.device) + bg_color
    if dirs.shape[0] == 0:
        return {"image": bg, "diffuse_image": bg.clone(), "specular_image": bg.clone(), "mask": mask, "depth": torch.zeros(N, device=rays_o.device)}
    out = shader.shade(normals.contiguous(), dirs.contiguous(), geo_feat, kappa_inv, env_rot_radian)
    put = lambda v: bg.masked_scatter(mask[:, None], v)
    depth = torch.zeros(N, device=rays_o.device).masked_scatter(mask, near[mask][:, 0])
    return {"image": put(out["c_diffuse"] + out["c_specular"]), "diffuse_image": put(out["c_diffuse"]), "specular_image": put(out["c_specular"]),
            "mask": mask, "depth": depth}


def _empty_results(model, prefix, bg_color, get_normal_image):
    """no ray reaches the sphere (reference sph_ray.py:58-68; weights_sum added so that chunks can be concatenated).  The optional
    images follow the conditions of the non-empty path (normal image in eval mode only, diffuse / specular only with use_diffuse), so
    that a chunk without a hit has the keys of a chunk with one"""
    N = bg_color.shape[0]
    vis = model.opt.visual_items
    out = {"image": bg_color.reshape(*prefix, 3), "depth": bg_color.new_zeros(*prefix),
           "normal_image": torch.zeros_like(bg_color).reshape(*prefix, 3) if (get_normal_image and not model.training) else None,
           "weights_sum": bg_color.new_zeros(N, 1), "empty": True}
    if model.opt.use_diffuse:
        if "diffuse" in vis:
            out["diffuse_image"] = bg_color.reshape(*prefix, 3)
        if "specular" in vis:
            out["specular_image"] = bg_color.reshape(*prefix, 3)
    if "roughness" in vis:
        out["roughness_image"] = torch.zeros_like(bg_color[..., :1]).reshape(*prefix, 1)
    return out


def run_sph(model, rays_o, rays_d, dt_gamma=0, bg_color=None, perturb=False, num_step=12, step_size=0.002, get_normal_image=False,
            use_specular_color=True, env_net_index=None, material=None, r_images=None, fused=True, **kwargs):
    """The reference's `run_sph` (sph_ray.py:34-221).  rays_o, rays_d [1,N,3].  Returns its result dict: image / depth (and, per
    `opt.visual_items`, diffuse_image / specular_image / roughness_image) shaped like the rays, normal_image or None, weights_sum
    [N,1], sigmas / sdfs [M,S] of the M hit rays (operator form only: the fused form never materialises sdf), and the training
    extras (surf_sdfs, relsdf ...).  `env_rot_radian` arrives in **kwargs and is ignored, as in the reference."""
    self = model
    opt = self.opt
    prefix = rays_o.shape[:-1]
    rays_o = rays_o.contiguous().view(-1, 3)
    rays_d = rays_d.contiguous().view(-1, 3)
    N, device = rays_o.shape[0], rays_o.device
    use_grad = (opt.debug or opt.backsdf_loss or opt.eikonal_loss or self.use_normal_with_mlp or self.use_n_dot_viewdir
                or self.use_reflected_dir)
    radius = opt.env_sph_radius
    bg_color = (torch.zeros(N, 3, device=device) + (1 if bg_color is None else bg_color)).reshape(N, 3)
    if fused and _sph_fused_ok(self, r_images):
        return _run_sph_fused(self, prefix, rays_o, rays_d, radius, bg_color, perturb, num_step, step_size, get_normal_image,
                              env_net_index or 0, material)
    nears, fars, mask = get_sphere_intersections(rays_o, rays_d, radius)
    if not mask.any():
        return _empty_results(self, prefix, bg_color, get_normal_image)
    nears_valid = nears[mask]                                                     # [M,1]
    z_radius = step_size * (num_step - 1) / 2
    z_vals = torch.linspace(-z_radius, z_radius, num_step, device=device)[None, :] + nears_valid      # [M,S]
    if perturb:
        z_vals = z_vals + (torch.rand_like(z_vals) - 0.5) * step_size
    dirs = rays_d[mask, None, :]                                                  # [M,1,3]
    xyzs = rays_o[mask, None, :] + dirs * z_vals[:, :, None]                      # [M,S,3]
    results = {}
    if use_grad or get_normal_image:
        xyzs.requires_grad_(True)
    with torch.enable_grad():
        sdfs, sigmas, geo_feats, normals, eik = self.forward_sigma(xyzs, use_sdf_sigma_grad=use_grad, material=material)
    roughness = getattr(self, "roughness", opt.default_roughness)
    with torch.set_grad_enabled(self.training):
        sigmas = self.density_scale * sigmas
        normals_enc, w_r_enc, n_dot_w_o, n_env_enc = self.get_color_mlp_extra_params(normals, dirs, roughness)
        if opt.train_renv:
            r_images = r_images[0, mask, None, :].expand(-1, num_step, -1)
        if not self.training:
            normals_enc = normals_enc.detach()
            w_r_enc, n_dot_w_o, n_env_enc = (None if t is None else t.detach() for t in (w_r_enc, n_dot_w_o, n_env_enc))
            geo_feats = geo_feats.detach()
        rgbs = self.forward_color(geo_feats, dirs, normals_enc, w_r_enc, n_dot_w_o, use_specular_color, env_net_index or 0,
                                  n_env_enc=n_env_enc, r_images=r_images, roughness=roughness)
        deltas = z_vals[..., 1:] - z_vals[..., :-1]
        deltas = torch.cat([deltas, step_size * torch.ones_like(deltas[..., :1])], dim=-1)
        alphas = 1 - torch.exp(-deltas * sigmas.squeeze(-1))
        alphas_shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
        weights = alphas * torch.cumprod(alphas_shifted, dim=-1)[..., :-1]
        weights_sum = weights.sum(dim=-1, keepdim=True)
    ori_z_vals = ((z_vals - nears_valid) / (fars.max() - nears_valid)).clamp(0, 1)
    depth = torch.zeros_like(nears).masked_scatter_(mask[..., None], torch.sum(weights * ori_z_vals, dim=-1))
    put3 = lambda v: bg_color.masked_scatter(mask[..., None], v + (1 - weights_sum) * bg_color[mask])
    image = put3(torch.sum(weights[..., None] * rgbs, dim=-2))
    normal_image = None
    if not self.training and get_normal_image:
        n = F.normalize(torch.sum(weights[..., None] * normals.detach(), dim=1), dim=-1)
        normal_image = torch.zeros(N, 3, device=device).masked_scatter_(mask[..., None], n).reshape(*prefix, 3)
    if opt.use_diffuse:
        if "diffuse" in opt.visual_items:
            results["diffuse_image"] = put3(torch.sum(weights[..., None] * self.c_diffuse, dim=-2)).reshape(*prefix, 3)
        if "specular" in opt.visual_items:
            results["specular_image"] = put3(torch.sum(weights[..., None] * self.c_specular, dim=-2)).reshape(*prefix, 3)
    if torch.is_tensor(getattr(self, "roughness", None)) and "roughness" in opt.visual_items:
        r_img = torch.sum(weights[..., None] * roughness, dim=-2)
        results["roughness_image"] = torch.zeros(N, 1, device=device).masked_scatter_(mask[..., None], r_img).reshape(*prefix, 1)
    results["weights_sum"] = torch.zeros_like(nears).masked_scatter_(mask[..., None], weights_sum)
    results["sigmas"], results["sdfs"] = sigmas, sdfs
    if opt.eikonal_loss:
        results["sdf_gradients"] = eik
    results["depth"], results["image"], results["normal_image"] = depth.reshape(*prefix), image.reshape(*prefix, 3), normal_image
    if self.training and getattr(opt, "sdf_loss_weight", 0) > 0:
        surf = rays_o[mask, None, :] + dirs * nears_valid[:, :, None]
        results["surf_sdfs"] = self.forward_sigma(surf, use_sdf_sigma_grad=False, material=material)[0]
    if opt.backsdf_loss:
        results["relsdf"] = sdfs[..., 1:] - sdfs[..., :-1]
        results["sdf_weights"], results["sdf_dist"] = weights[..., :-1], deltas[..., :-1]
    return results


def _sph_fused_ok(model, r_images) -> bool:
    """the four-launch form: a GPU model in eval mode whose network shapes the fused kernels are built for"""
    return (not model.training and r_images is None and next(model.parameters()).is_cuda and not model.opt.train_renv
            and getattr(model, "supports_fused_sph", lambda: False)())


def _run_sph_fused(model, prefix, rays_o, rays_d, radius, bg_color, perturb, num_step, step_size, get_normal_image, env_net_index, material):
    opt = model.opt
    N, device = rays_o.shape[0], rays_o.device
    fr = model.fused_sph_renderer(env_net_index, material)
    rays_o, rays_d = rays_o.float().contiguous(), rays_d.float().contiguous()
    nears, fars, mask = fr.sphere_intersections(rays_o, rays_d, radius)
    hit_rays = torch.nonzero(mask).squeeze(-1).to(torch.int32)                    # (the one host round trip: M sizes the sample arrays)
    M = int(hit_rays.shape[0])
    if M == 0:
        return _empty_results(model, prefix, bg_color, get_normal_image)
    hit_slot = torch.full((N,), -1, dtype=torch.int32, device=device)
    hit_slot[hit_rays.long()] = torch.arange(M, dtype=torch.int32, device=device)
    z_radius = step_size * (num_step - 1) / 2
    z_offsets = torch.linspace(-z_radius, z_radius, num_step, device=device)
    noise = torch.rand(M, num_step, device=device) if perturb else None
    near1 = nears
    xyz, dirs, z_vals = fr.shell_samples(rays_o, rays_d, hit_rays, near1, z_offsets, step_size, noise)
    geo = fr.geometry_eval(xyz, want=("sigma", "normal", "geo_feat", "roughness"))
    shaded = fr.shade(geo["normal"], dirs, geo["geo_feat"], geo["roughness"], None)
    want = tuple(k for k, on in (("normal_image", get_normal_image), ("diffuse_image", opt.use_diffuse and "diffuse" in opt.visual_items),
                                 ("specular_image", opt.use_diffuse and "specular" in opt.visual_items),
                                 ("roughness_image", "roughness" in opt.visual_items)) if on)
    out = fr.composite_shell(geo["sigma"], z_vals, shaded["c_diffuse"], shaded["c_specular"], geo["normal"], geo["roughness"], hit_slot, near1,
                             fars.max().reshape(1), bg_color.contiguous(), step_size, want=want)
    results = {"image": out["image"].reshape(*prefix, 3), "depth": out["depth"].reshape(*prefix), "weights_sum": out["weights_sum"].reshape(N, 1),
               "normal_image": out["normal_image"].reshape(*prefix, 3) if get_normal_image else None,
               "sigmas": geo["sigma"].reshape(num_step, M).t()}
    for k in ("diffuse_image", "specular_image"):
        if k in out:
            results[k] = out[k].reshape(*prefix, 3)
    if "roughness_image" in out:
        results["roughness_image"] = out["roughness_image"].reshape(*prefix, 1)
    return results
