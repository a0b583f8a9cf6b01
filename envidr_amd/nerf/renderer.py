"""`NeRFRenderer`: scene box + occupancy-grid state + `render()` dispatch, on the HIP operators.

Mirrors the reference's nerf/renderer.py for the cuda_ray path: same constructor arguments, buffers
(`aabb_train`, `aabb_infer`, `density_grid`, `density_bitfield`, `step_counter`), helper methods
(`get_color_mlp_extra_params`, `compute_normal`) and the `render()` signature / result dict
(renderer.py:364-530).  Inference renders go through the fused persistent kernel when the model
configuration is one it implements (`fused=True`, default), otherwise through the operator-by-
operator loop of render_func/cuda_ray.py, which is the reference's loop on HIP operators.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import raymarching
from . import render_func
from .utils import rot_theta

SQRT3 = 3 ** 0.5


def reflect_dir(w_o: torch.Tensor, normals: torch.Tensor) -> torch.Tensor:
    """mirror w_o (surface -> camera, unit) about the unit normal: 2 (n . w_o) n - w_o"""
    return 2 * torch.sum(w_o * normals, dim=-1, keepdim=True) * normals - w_o


class NeRFRenderer(nn.Module):
    def __init__(self, bound=1, cuda_ray=False, density_scale=1, min_near=0.2, density_thresh=0.01, bg_radius=-1,
                 use_sdf=True, opt=None, env_opt=None, **kwargs):
        super().__init__()
        self.opt, self.env_opt = opt, env_opt
        self.bound = bound
        self.cascade = 1 + math.ceil(math.log2(bound))
        self.grid_size = 128
        self.density_scale = density_scale
        self.min_near = min_near
        self.density_thresh = density_thresh
        self.bg_radius = bg_radius
        marching_aabb = list(getattr(opt, "marching_aabb", []) or [])
        if len(marching_aabb) == 6:
            aabb = (torch.tensor(marching_aabb, dtype=torch.float32) * opt.scale).clamp(-bound, bound)
        else:
            aabb = torch.tensor([-bound] * 3 + [bound] * 3, dtype=torch.float32)
        self.register_buffer("aabb_train", aabb)
        self.register_buffer("aabb_infer", aabb.clone())
        self.obj_aabb = None
        self.use_sdf = use_sdf
        self.use_normal_with_mlp = opt.normal_with_mlp
        self.use_reflected_dir = opt.use_reflected_dir
        self.use_n_dot_viewdir = opt.use_n_dot_viewdir
        self.cuda_ray = cuda_ray
        if cuda_ray:
            self.register_buffer("density_grid", torch.zeros(self.cascade, self.grid_size ** 3))
            self.register_buffer("density_bitfield", torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8))
            self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))
            self.mean_density = 0
            self.iter_density = 0
            self.mean_count = 0
            self.local_step = 0
        self._fused = None
        self._fused_key = None

    # ---- per-sample helpers (reference renderer.py:147-198) -------------------------------------
    def get_color_mlp_extra_params(self, normals, dirs, roughness=0, env_rot_radian=None):
        if normals is None:
            return None, None, None, None
        normals_enc = self.encoder_normal(normals) if self.use_normal_with_mlp else None
        w_o = -dirs
        rot = None
        if env_rot_radian is not None:
            rot = torch.from_numpy(rot_theta(env_rot_radian)[:3, :3]).float().to(normals.device)
        w_r_enc = None
        if self.use_reflected_dir and not self.opt.diffuse_only:
            w_r = reflect_dir(w_o, normals)
            if rot is not None:
                w_r = w_r @ rot
            w_r_enc = self.encoder_refdir(w_r, roughness=roughness) * self.opt.light_intensity_scale
        n_dot_w_o = torch.sum(normals * w_o, dim=-1, keepdim=True) if self.use_n_dot_viewdir else None
        n_env_enc = None
        if self.opt.diffuse_with_env:
            n_env = normals @ rot if rot is not None else normals
            enc = self.diffuse_encoder_refdir if self.opt.split_diffuse_env else self.encoder_refdir
            n_env_enc = enc(n_env, roughness=self.opt.diffuse_kappa_inv) * self.opt.light_intensity_scale
        return normals_enc, w_r_enc, n_dot_w_o, n_env_enc

    def compute_normal(self, sdf_or_sigma, xyzs, get_eikonal_sdf_gradient=False):
        """normal = normalize(d sdf / d xyz) through autograd (the fused kernel does this analytically)"""
        grad = torch.autograd.grad(sdf_or_sigma, xyzs, torch.ones_like(sdf_or_sigma), retain_graph=True, create_graph=True)[0]
        if not self.use_sdf:
            grad = -grad
        eikonal = grad if get_eikonal_sdf_gradient else None
        normals = grad.detach() if self.opt.detach_normal else grad
        normals = F.normalize(normals, dim=-1, eps=1e-10)
        if self.opt.normal_anneal_ratio < 1:
            r = self.opt.normal_anneal_ratio
            normals = F.normalize(normals * r + (1 - r) * F.normalize(xyzs.detach(), dim=-1, eps=1e-10), dim=-1, eps=1e-10)
        return normals, eikonal

    # ---- render ----------------------------------------------------------------------------------
    def fused_renderer(self):
        """the fused persistent-kernel renderer for this model, (re)built when parameters change
        (call `invalidate_fused()` after loading / editing weights)."""
        if self._fused is None:
            self._fused = self._build_fused()
        return self._fused

    def invalidate_fused(self):
        self._fused = None

    def _build_fused(self):
        raise NotImplementedError

    def supports_fused(self, **kwargs) -> bool:
        return False

    def render(self, rays_o, rays_d, staged=False, max_ray_batch=4096, get_normal_image=False, use_specular_color=True,
               env_net_index=None, material=None, r_images=None, env_rot_radian=None, fused=True, **kwargs):
        """rays_o, rays_d: [B, N, 3] (B == 1).  Returns the reference's result dict: image [B,N,3],
        depth [B,N], weights_sum [B,N] and, per configuration, normal_image / diffuse_image /
        specular_image / roughness_image."""
        if not self.cuda_ray:
            raise NotImplementedError("only the cuda_ray render path is implemented (reference non-cuda paths are out of scope)")
        kwargs["material"] = material
        if self.opt.indir_ref:
            return self._render_indirect(rays_o, rays_d, get_normal_image, use_specular_color, env_net_index, env_rot_radian,
                                         fused, **kwargs)
        results = self._run(rays_o, rays_d, get_normal_image=get_normal_image, use_specular_color=use_specular_color,
                            env_net_index=env_net_index, r_images=r_images, env_rot_radian=env_rot_radian, fused=fused, **kwargs)
        if "weights_sum" in results and get_normal_image and results.get("normal_image") is not None:
            ws = results["weights_sum"][..., None]
            results["normal_image"] = results["normal_image"] * ws + (1 - ws)
        return results

    def _run(self, rays_o, rays_d, fused=True, **kw):
        batch = self.opt.max_ray_batch_cuda
        N = rays_o.shape[1]
        if batch is None or batch <= 0 or N <= batch:
            return render_func.run_cuda(self, rays_o, rays_d, fused=fused, **kw)
        chunks = [render_func.run_cuda(self, rays_o[:, i:i + batch], rays_d[:, i:i + batch], fused=fused, **kw)
                  for i in range(0, N, batch)]
        out = {}
        for k in chunks[0]:
            if chunks[0][k] is None:
                out[k] = None
            else:
                out[k] = torch.cat([c[k] for c in chunks], 0 if k in ("diffuse_image", "specular_image", "roughness_image") else 1)
        return out

    def _render_indirect(self, rays_o, rays_d, get_normal_image, use_specular_color, env_net_index, env_rot_radian, fused, **kwargs):
        """three passes (reference renderer.py:437-513): geometry only -> reflected rays -> main pass
        with the reflected radiance fed to the renv branch."""
        dt = 2 * SQRT3 / self.opt.indir_max_steps
        geo = self._run(rays_o, rays_d, get_normal_image=get_normal_image, main_pass=False, geometry_only=True,
                        env_rot_radian=env_rot_radian, fused=False, **kwargs)
        normals = geo["normal_image"]
        depth = geo["depth"].squeeze() - dt
        ws = geo["weights_sum"].squeeze()
        ref_mask = (depth != 0) & (ws > 0.9)
        ray_mask = (depth != 0) & (ws > 0.3)
        ref_o = rays_o + depth[None, ..., None] * rays_d
        ref_d = reflect_dir(-rays_d, normals)
        saved_bg, saved_near = kwargs.get("bg_color"), self.min_near
        bg = 0 if saved_bg is None else saved_bg
        self.min_near = dt * 2
        kw2 = dict(kwargs, bg_color=0, max_steps=self.opt.indir_max_steps, early_stop_steps=self.opt.indir_early_stop_steps,
                   force_all_rays=True)
        try:
            ref = self._run(ref_o[:, ref_mask, :], ref_d[:, ref_mask, :], get_normal_image=get_normal_image,
                            use_specular_color=use_specular_color, env_net_index=env_net_index, main_pass=False,
                            bg_sphere=False, env_rot_radian=env_rot_radian, fused=fused, **kw2)
        finally:
            self.min_near = saved_near
        ref_image = torch.cat([ref["image"], ref["weights_sum"][..., None]], -1)
        ref2ray = ref_mask[ray_mask]
        r_images = ref_image.new_zeros(1, ref2ray.shape[0], 4).masked_scatter(ref2ray[None, :, None], ref_image)
        kw3 = dict(kwargs, bg_color=0)
        res = self._run(rays_o[:, ray_mask, :], rays_d[:, ray_mask, :], get_normal_image=get_normal_image,
                        use_specular_color=use_specular_color, env_net_index=env_net_index, main_pass=True, r_images=r_images,
                        bg_sphere=False, env_rot_radian=env_rot_radian, fused=False, **kw3)
        res["normal_image"] = normals
        res["depth"] = depth[None, :]
        for k in ("image", "specular_image", "diffuse_image", "roughness_image"):
            if k in res and res[k] is not None:
                v = res[k].reshape(1, -1, res[k].shape[-1]) if res[k].dim() == 2 else res[k]
                res[k] = normals.new_zeros(*normals.shape[:-1], v.shape[-1]).masked_scatter(ray_mask[None, :, None], v)
        wsum = normals.new_zeros(*normals.shape[:-1], 1).masked_scatter(ray_mask[None, :, None], res["weights_sum"][..., None])
        res["image"] = (torch.zeros_like(normals) + bg) * (1 - wsum) + res["image"]
        res["weights_sum"] = wsum.squeeze(-1)
        if get_normal_image:
            w = res["weights_sum"][..., None]
            res["normal_image"] = res["normal_image"] * w + (1 - w)
        return res
