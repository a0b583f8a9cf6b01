"""Surface rendering of the environment-lit sphere: BASELINE configs[0] (the reference's `demo.ipynb`, "Rendering Steps"
cell) and the ray / sphere intersection its env-sphere mode is built on (nerf/render_func/sph_ray.py:18-32).

The reference's `run_sph` volume-renders 12 samples around the sphere with its Trainer-side options; that loop is out of
scope (semantics only).  What is on the path is the notebook's form -- one surface sample per hit
ray, shaded by IDE x2 + environment MLP x2 + diffuse / specular heads -- which is exactly `envidr_shade_samples`
(`FusedShader.shade`, csrc/fused_render.hip k_shade_samples): the geometry is analytic, the shading runs on the HIP kernel.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def get_sphere_intersections(rays_o: torch.Tensor, rays_d: torch.Tensor, r: float = 1.0):
    """near / far ray parameters of the sphere |x| = r and the mask of rays that reach it; rays_o, rays_d [N,3] (unit
    directions) -> near [N,1], far [N,1], mask [N].  Reference sph_ray.py:18-32: discriminant clamped at 0 under the root,
    a ray counts as a hit from a discriminant of -1e-4 (grazing rays)."""
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
