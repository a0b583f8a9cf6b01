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

import numpy as np
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
        # options that select code of the reference this package does not carry are refused here, never ignored:
        # error_bound_sample -> the VolSDF sampler (reference renderer.py:373-375); unwrap_env_sphere -> main_nerf.py's environment-map
        # export; plot_roughness -> matplotlib debugging inside forward_geometry (network.py:336,402,452).  (env_sph_mode /
        # render_env_on_sphere select run_sph, renderer.py:376-377: render_func/sph_ray.py here, since round 5.)
        for name in ("error_bound_sample", "unwrap_env_sphere", "plot_roughness"):
            if getattr(opt, name, False):
                raise NotImplementedError(f"opt.{name} selects a part of the reference outside the render hot path (DESIGN.md section 7)")
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
        # reference renderer.py:91-97: the object's box (opt.obj_aabb * scale, clamped to the bound cube).  It masks the
        # reflected rays of indirect rendering (renderer.py:458-460) and the samples of the relative-sdf term (cuda_ray.py:191-192)
        obj_aabb = getattr(opt, "obj_aabb", None)
        if obj_aabb is not None and len(obj_aabb) > 0:
            if len(obj_aabb) != 6:
                raise ValueError(f"obj_aabb takes 6 numbers (p min, p max), got {len(obj_aabb)}")
            self.register_buffer("obj_aabb", (torch.tensor(list(obj_aabb), dtype=torch.float32) * opt.scale).clamp(min=-bound, max=bound),
                                 persistent=False)
        else:
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
        self._fused_sph = {}          # env-sphere mode: one fused renderer per environment MLP (env_net_index)

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
        # (only the position gradient of this pass is returned: the hash encoder skips the table gradient it would otherwise form and
        #  autograd.grad would throw away -- hashencoder.input_gradient_only; the pass stays twice differentiable)
        from ..hashencoder.hashgrid import input_gradient_only
        with input_gradient_only():
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

    def reset_extra_state(self):
        """empty density grid and step statistics (reference renderer.py:131-141)"""
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.mean_density = 0
        self.iter_density = 0
        self.step_counter.zero_()
        self.mean_count = 0
        self.local_step = 0

    # ---- occupancy-grid maintenance (reference renderer.py:200-359) ------------------------------
    def _cell_blocks(self, S):
        """the H^3 cell lattice in S^3 blocks: (coords [n,3] int32, morton indices [n] int64) per block,
        block order x-major like the reference's nested loops (it fixes the order random jitter is drawn in)"""
        dev = self.density_bitfield.device
        axis = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
        for xs in axis:
            for ys in axis:
                for zs in axis:
                    coords = torch.stack(torch.meshgrid(xs, ys, zs, indexing="ij"), dim=-1).reshape(-1, 3)
                    yield coords, raymarching.morton3D(coords).long()

    def _cell_centres(self, coords):
        """2 c / (H - 1) - 1 per axis, from a host-computed (IEEE fp32) table: device division kernels are not
        guaranteed correctly rounded, and one ulp in a position is visible through a fine hash level"""
        c = np.arange(self.grid_size, dtype=np.float32)
        table = torch.from_numpy(np.float32(2) * c / np.float32(self.grid_size - 1) - np.float32(1)).to(coords.device)
        return table[coords.long()]

    def _cascade_extent(self, cas):
        bound = min(2 ** cas, self.bound)
        return bound, bound / self.grid_size            # (half extent of the cascade cube, half a cell)

    # Random draws of the grid update.  `self.grid_rng` (an object with rand(shape) / randint(high, shape))
    # replaces the device generator; the parity test uses it to replay the reference's CPU random stream.
    def _unit_noise(self, ref):
        rng = getattr(self, "grid_rng", None)
        if rng is not None:
            return rng.rand(tuple(ref.shape)).to(ref.device, ref.dtype)
        return torch.rand_like(ref)

    def _randint(self, high, shape, dev):
        rng = getattr(self, "grid_rng", None)
        if rng is not None:
            return rng.randint(high, tuple(shape)).to(dev)
        return torch.randint(0, high, tuple(shape), device=dev)

    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        """cells no training camera sees get density -1 (never occupied, never updated).
        poses [B,4,4] camera-to-world, intrinsic (fx, fy, cx, cy)."""
        if not self.cuda_ray:
            return
        dev = self.density_bitfield.device
        poses = torch.as_tensor(poses, dtype=torch.float32).to(dev)
        fx, fy, cx, cy = intrinsic
        seen = torch.zeros_like(self.density_grid)
        R, t = poses[:, :3, :3], poses[:, :3, 3]
        for coords, indices in self._cell_blocks(S):
            centres = self._cell_centres(coords).unsqueeze(0)                                   # [1,n,3] in [-1,1]
            for cas in range(self.cascade):
                bound, half_cell = self._cascade_extent(cas)
                world = centres * (bound - half_cell)
                for b0 in range(0, poses.shape[0], S):
                    cam = (world - t[b0:b0 + S].unsqueeze(1)) @ R[b0:b0 + S]                    # world -> camera (R is c2w)
                    z = cam[:, :, 2]
                    inside = (z > 0) & (cam[:, :, 0].abs() < cx / fx * z + half_cell * 2) \
                        & (cam[:, :, 1].abs() < cy / fy * z + half_cell * 2)
                    seen[cas, indices] += inside.sum(0).reshape(-1)
        self.density_grid[seen == 0] = -1
        return int((seen == 0).sum())

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128, full_update=False):
        """refresh density_grid (jittered density query per cell, decayed running max) and re-pack the
        occupancy bitfield; also folds the marcher's step counter into mean_count."""
        if not self.cuda_ray:
            return
        dev = self.density_bitfield.device
        fresh = -torch.ones_like(self.density_grid)

        def query(cas, coords, indices):
            bound, half_cell = self._cascade_extent(cas)
            xyzs = self._cell_centres(coords) * (bound - half_cell)
            xyzs += (self._unit_noise(xyzs) * 2 - 1) * half_cell
            sigmas = self.density(xyzs)["sigma"].reshape(-1).detach()
            sigmas *= self.density_scale
            fresh[cas, indices] = sigmas

        if self.iter_density < 16 or full_update:
            for coords, indices in self._cell_blocks(S):
                for cas in range(self.cascade):
                    query(cas, coords, indices)
        else:
            # a quarter of the cells at random + as many draws (with repetition) from the occupied ones
            n = self.grid_size ** 3 // 4
            for cas in range(self.cascade):
                coords = self._randint(self.grid_size, (n, 3), dev)
                indices = raymarching.morton3D(coords).long()
                occupied = torch.nonzero(self.density_grid[cas] > 0).squeeze(-1)
                occupied = occupied[self._randint(occupied.shape[0], (n,), dev)]
                query(cas, torch.cat([coords, raymarching.morton3D_invert(occupied)], 0), torch.cat([indices, occupied], 0))

        live = (self.density_grid >= 0) & (fresh >= 0)
        self.density_grid[live] = torch.maximum(self.density_grid[live] * decay, fresh[live])
        self.mean_density = torch.mean(self.density_grid.clamp(min=0)).item()       # -1 cells count as empty
        self.iter_density += 1
        self.density_bitfield = raymarching.packbits(self.density_grid, min(self.mean_density, self.density_thresh),
                                                     self.density_bitfield)
        steps = min(16, self.local_step)
        if steps > 0:
            self.mean_count = int(self.step_counter[:steps, 0].sum().item() / steps)
        self.local_step = 0
        self.invalidate_fused()

    # ---- render ----------------------------------------------------------------------------------
    def fused_renderer(self):
        """the fused persistent-kernel renderer for this model, (re)built when parameters change
        (call `invalidate_fused()` after loading / editing weights)."""
        if self._fused is None:
            self._fused = self._build_fused()
        return self._fused

    def invalidate_fused(self):
        self._fused = None
        self._fused_sph = {}

    def _build_fused(self):
        raise NotImplementedError

    def supports_fused(self, **kwargs) -> bool:
        return False

    def render(self, rays_o, rays_d, staged=False, max_ray_batch=4096, get_normal_image=False, use_specular_color=True,
               env_net_index=None, material=None, r_images=None, env_rot_radian=None, fused=True, **kwargs):
        """rays_o, rays_d: [B, N, 3] (any B: the leading shape is flattened and restored like the reference's `prefix`,
        cuda_ray.py:30-33).  Returns the reference's result dict: image [B,N,3],
        depth [B,N], weights_sum [B,N] and, per configuration, normal_image / diffuse_image /
        specular_image / roughness_image."""
        if self.opt.env_sph_mode or getattr(self.opt, "render_env_on_sphere", False):
            return self._render_sph(rays_o, rays_d, staged, max_ray_batch, get_normal_image, use_specular_color, env_net_index, material,
                                    r_images, env_rot_radian, fused, **kwargs)
        if self.opt.error_bound_sample:
            raise NotImplementedError("the VolSDF sampler (reference run_volsdf) is out of scope")
        if not self.cuda_ray:
            return self._render_plain(rays_o, rays_d, staged, max_ray_batch, get_normal_image, use_specular_color, env_net_index, material,
                                      r_images, env_rot_radian, **kwargs)
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

    def _render_plain(self, rays_o, rays_d, staged, max_ray_batch, get_normal_image, use_specular_color, env_net_index, material, r_images,
                      env_rot_radian, **kwargs):
        """cuda_ray = False: `_run = render_func.run` (reference renderer.py:368-371) under render()'s two chunking forms -- staged chunks of
        max_ray_batch rays written into preallocated frames (:381-423; its normal frame is torch.empty where a chunk returns none) or
        max_ray_batch_cuda (:425-436) -- and the normal-image blend of :539-540."""
        kwargs["material"] = material
        B, N = rays_o.shape[:2]
        call = lambda o, d, r: render_func.run(self, o, d, get_normal_image=get_normal_image, use_specular_color=use_specular_color,
                                               env_net_index=env_net_index, r_images=r, env_rot_radian=env_rot_radian, **kwargs)
        if not staged:
            # one call (the reference's max_ray_batch_cuda chunks cannot be joined for this function: it concatenates the 1-D weights_sum
            # along dimension 1); the blend of renderer.py:529-530 on the function's own [N,3] normal image
            results = call(rays_o, rays_d, r_images)
            if get_normal_image and results.get("normal_image") is not None:
                ws = results["weights_sum"][..., None]
                results["normal_image"] = results["normal_image"] * ws + (1 - ws)
            return results
        parts = [[call(rays_o[b:b + 1, i:i + max_ray_batch], rays_d[b:b + 1, i:i + max_ray_batch],
                       None if r_images is None else r_images[b:b + 1, i:i + max_ray_batch]) for i in range(0, N, max_ray_batch)] for b in range(B)]
        join = lambda k, shape: torch.cat([torch.cat([p[k].reshape(1, -1, *shape) for p in row], 1) for row in parts], 0)
        results = {"depth": join("depth", ()), "image": join("image", (3,))}
        # (the staged frames carry no weights_sum, so the reference does not blend the normal image here; its normal frame is torch.empty
        #  where no chunk returned one -- zeros here.  The reference itself cannot finish a staged render with this function: it then reads
        #  results_['roughness_image'] & co for every name in visual_items, which `run` never returns, renderer.py:407-412)
        results["normal_image"] = join("normal_image", (3,)) if parts[0][0]["normal_image"] is not None else torch.zeros(B, N, 3, device=rays_o.device)
        return results

    def _render_sph(self, rays_o, rays_d, staged, max_ray_batch, get_normal_image, use_specular_color, env_net_index, material, r_images,
                    env_rot_radian, fused, **kwargs):
        """env-sphere mode: `_run = render_func.run_sph` (reference renderer.py:376-377) under render()'s two chunking forms
        (:381-423 staged chunks of max_ray_batch rays; :425-436 max_ray_batch_cuda) and its normal-image blend (:539-540)."""
        kwargs["material"] = material
        B, N = rays_o.shape[:2]
        if B != 1:
            raise ValueError("env-sphere mode renders one view per call (the reference's run_sph assumes B == 1)")
        batch = max_ray_batch if staged else (self.opt.max_ray_batch_cuda if (self.opt.max_ray_batch_cuda or 0) > 0 else N)
        call = lambda o, d, r: render_func.run_sph(self, o, d, get_normal_image=get_normal_image, use_specular_color=use_specular_color,
                                                   env_net_index=env_net_index, r_images=r, env_rot_radian=env_rot_radian, fused=fused, **kwargs)
        if N <= batch:
            results = call(rays_o, rays_d, r_images)
        else:
            # every chunk is a run_sph call of its own, as in the reference: the depth of a chunk is normalised with the largest
            # `far` of THAT chunk (sph_ray.py:112)
            parts = [call(rays_o[:, i:i + batch], rays_d[:, i:i + batch], None if r_images is None else r_images[:, i:i + batch])
                     for i in range(0, N, batch)]
            results = {}
            hit = [p for p in parts if not p.get("empty")]
            for k in ("image", "depth", "weights_sum", "normal_image", "diffuse_image", "specular_image", "roughness_image"):
                # a key is part of the frame if the chunks that hit the sphere carry it (all of them do or none); a chunk without a hit
                # contributes its background / zeros (its own entry where it has one)
                if not any(p.get(k) is not None for p in (hit or parts)):
                    continue
                vals = []
                for p in parts:
                    v = p.get(k)
                    if v is None:
                        img = p["image"]
                        v = img if k in ("diffuse_image", "specular_image") else img.new_zeros(*img.shape[:-1], 1 if k == "roughness_image" else 3)
                    vals.append(v)
                results[k] = torch.cat(vals, 0 if k == "weights_sum" else 1)
        if "weights_sum" in results and get_normal_image and results.get("normal_image") is not None:
            # the reference multiplies [B,N,3] by weights_sum[..., None] with weights_sum [N,1] here, which broadcasts to [N,N,3] (an
            # accident of run_sph returning an un-reshaped weights_sum); the per-ray blend it means -- its diagonal -- is what is returned
            ws = results["weights_sum"].reshape(*results["normal_image"].shape[:-1], 1)
            results["normal_image"] = results["normal_image"] * ws + (1 - ws)
        return results

    def _run(self, rays_o, rays_d, fused=True, **kw):
        batch = self.opt.max_ray_batch_cuda
        N = rays_o.shape[1]
        if batch is None or batch <= 0 or N <= batch:
            return render_func.run_cuda(self, rays_o, rays_d, fused=fused, **kw)
        # per-ray inputs travel with their chunk: r_images is indexed by chunk-local ray id inside run_cuda (the reference
        # chunks outside its three-pass block, renderer.py:416-435, so its r_images is always aligned with the chunk)
        r_images = kw.pop("r_images", None)
        chunks = [render_func.run_cuda(self, rays_o[:, i:i + batch], rays_d[:, i:i + batch], fused=fused,
                                       r_images=None if r_images is None else r_images[:, i:i + batch].contiguous(), **kw)
                  for i in range(0, N, batch)]
        out = {}
        for k in chunks[0]:
            if chunks[0][k] is None:
                out[k] = None
            elif torch.is_tensor(chunks[0][k]):
                out[k] = torch.cat([c[k] for c in chunks], 0 if k in ("diffuse_image", "specular_image", "roughness_image") else 1)
            # anything else is chunk-local state of the fused path ("_frame": the raw frame whose records a following pass over the SAME
            # rays could shade again) -- it does not describe the concatenated rays and is dropped
        return out

    def _render_indirect_masked(self, rays_o, rays_d, get_normal_image, use_specular_color, env_net_index, env_rot_radian, **kwargs):
        """The same three passes on the fused geometry pipeline, with ray MASKS instead of boolean-mask gathers: every pass runs
        over all N rays (masked-out rays cost a byte load), r_images is indexed by ray id directly, and nothing between the
        passes needs a count on the host -- no nonzero, no masked_scatter, no synchronisation (the reference gathers the
        selected rays, renderer.py:455-470,490-512, which costs three host round trips per frame)."""
        from ..fused import FrameOverflow
        for attempt in range(3):
            try:
                # the three passes are enqueued back to back; whether they fitted their buffers is looked at once, at the end
                res = self._indirect_passes(rays_o, rays_d, get_normal_image, use_specular_color, env_net_index, env_rot_radian, **kwargs)
                self.fused_renderer().check_frames()
                return res
            except FrameOverflow:
                if attempt == 2:
                    raise
        raise AssertionError("unreachable")

    def _indirect_passes(self, rays_o, rays_d, get_normal_image, use_specular_color, env_net_index, env_rot_radian, **kwargs):
        dt = 2 * SQRT3 / self.opt.indir_max_steps
        kwargs = dict(kwargs, wait=False)
        geo = self._run(rays_o, rays_d, get_normal_image=get_normal_image, main_pass=False, geometry_only=True,
                        env_rot_radian=env_rot_radian, fused=True, frame_tag="indirect-geometry", frame_buffers="indirect-primary", **kwargs)
        normals = geo["normal_image"]                       # [1,N,3]
        depth = geo["depth"] - dt                           # [1,N]
        ws = geo["weights_sum"]
        ref_mask = (depth != 0) & (ws > 0.9)
        ray_mask = (depth != 0) & (ws > 0.3)
        ref_o = rays_o + depth[..., None] * rays_d
        ref_d = reflect_dir(-rays_d, normals)
        if self.obj_aabb is not None:                       # reflected rays start inside the object's box (reference renderer.py:458-460)
            ref_mask = ref_mask & (ref_o > self.obj_aabb[:3]).all(-1) & (ref_o < self.obj_aabb[3:]).all(-1)
        saved_bg, saved_near = kwargs.get("bg_color"), self.min_near
        bg = 0 if saved_bg is None else saved_bg
        if geo.get("sphere_bg") is not None:
            bg = geo["sphere_bg"].reshape(1, -1, 3)        # the background model's colours, computed by the first pass (reference renderer.py:466-467)
        self.min_near = dt * 2
        kw2 = dict(kwargs, bg_color=0, max_steps=self.opt.indir_max_steps, early_stop_steps=self.opt.indir_early_stop_steps,
                   force_all_rays=True)
        try:
            ref = self._run(ref_o, ref_d, get_normal_image=get_normal_image, use_specular_color=use_specular_color,
                            env_net_index=env_net_index, main_pass=False, grad_ray=getattr(self.opt, "grad_rays", False), bg_sphere=False,
                            env_rot_radian=env_rot_radian, fused=True, ray_mask=ref_mask, frame_tag="indirect-reflected", **kw2)
        finally:
            self.min_near = saved_near
        # reflected radiance + visibility per primary ray; zero where no reflected ray was traced (the reference's new_zeros +
        # masked_scatter, renderer.py:483-486)
        r_images = torch.cat([ref["image"], ref["weights_sum"][..., None]], -1) * ref_mask[..., None]
        kw3 = dict(kwargs, bg_color=0)
        # The main pass marches the same rays with the same parameters as the first: same samples, same compositing weights.  The
        # first pass's records are still in their buffer set (the reflected pass has its own), so the main pass only shades and
        # composites them again -- with the reflected radiance this time -- instead of marching and evaluating the SDF network a
        # second time; ray_mask zeroes the rays the reference would not have gathered (renderer.py:490-492).  (Chunked renders
        # -- max_ray_batch_cuda -- do not take this form: _render_indirect sends them through the gather path.)
        res = self._run(rays_o, rays_d, get_normal_image=get_normal_image, use_specular_color=use_specular_color,
                        env_net_index=env_net_index, main_pass=True, r_images=r_images, bg_sphere=False, env_rot_radian=env_rot_radian,
                        fused=True, ray_mask=ray_mask, frame_tag="indirect-main", frame_buffers="indirect-primary",
                        reuse_geometry=geo.get("_frame"), **kw3)
        res["normal_image"] = normals
        res["depth"] = depth
        for k in ("specular_image", "diffuse_image", "roughness_image"):
            if k in res and res[k] is not None:
                res[k] = res[k].reshape(1, -1, res[k].shape[-1])
        wsum = res["weights_sum"][..., None]
        res["image"] = (torch.zeros_like(normals) + bg) * (1 - wsum) + res["image"]
        if get_normal_image:
            res["normal_image"] = res["normal_image"] * wsum + (1 - wsum)
        return res

    def _render_indirect(self, rays_o, rays_d, get_normal_image, use_specular_color, env_net_index, env_rot_radian, fused, **kwargs):
        """three passes (reference renderer.py:437-513): geometry only -> reflected rays -> main pass
        with the reflected radiance fed to the renv branch."""
        batch = self.opt.max_ray_batch_cuda
        # the ray-mask form needs every one of its three passes to take run_cuda's fused branch (the operator loop has no
        # ray_mask): ask run_cuda's own eligibility test, with each pass's arguments
        gate = dict(fused=fused, ray_depth=kwargs.get("ray_depth"), perturb=kwargs.get("perturb", False))
        if (kwargs.get("two_phase") is not False and (batch is None or batch <= 0 or rays_o.shape[1] <= batch)
                and not torch.is_tensor(kwargs.get("bg_color"))
                and render_func.fused_eligible(self, geometry_only=True, max_steps=kwargs.get("max_steps", 1024), **gate)
                and render_func.fused_eligible(self, max_steps=self.opt.indir_max_steps, **gate)
                and render_func.fused_eligible(self, r_images=rays_o.new_zeros(1, 1, 4), max_steps=kwargs.get("max_steps", 1024), **gate)):
            return self._render_indirect_masked(rays_o, rays_d, get_normal_image, use_specular_color, env_net_index, env_rot_radian, **kwargs)
        dt = 2 * SQRT3 / self.opt.indir_max_steps
        geo = self._run(rays_o, rays_d, get_normal_image=get_normal_image, main_pass=False, geometry_only=True,
                        env_rot_radian=env_rot_radian, fused=fused, **kwargs)
        normals = geo["normal_image"]
        depth = geo["depth"].squeeze() - dt
        ws = geo["weights_sum"].squeeze()
        ref_mask = (depth != 0) & (ws > 0.9)
        ray_mask = (depth != 0) & (ws > 0.3)
        ref_o = rays_o + depth[None, ..., None] * rays_d
        ref_d = reflect_dir(-rays_d, normals)
        if self.obj_aabb is not None:                       # reference renderer.py:458-460
            ref_mask = ref_mask & (ref_o[0] > self.obj_aabb[:3]).all(-1) & (ref_o[0] < self.obj_aabb[3:]).all(-1)
        saved_bg, saved_near = kwargs.get("bg_color"), self.min_near
        bg = 0 if saved_bg is None else saved_bg
        if geo.get("sphere_bg") is not None:
            bg = geo["sphere_bg"].reshape(1, -1, 3)        # the background model's colours, computed by the first pass (reference renderer.py:466-467)
        self.min_near = dt * 2
        kw2 = dict(kwargs, bg_color=0, max_steps=self.opt.indir_max_steps, early_stop_steps=self.opt.indir_early_stop_steps,
                   force_all_rays=True)
        try:
            ref = self._run(ref_o[:, ref_mask, :], ref_d[:, ref_mask, :], get_normal_image=get_normal_image,
                            use_specular_color=use_specular_color, env_net_index=env_net_index, main_pass=False,
                            grad_ray=getattr(self.opt, "grad_rays", False), bg_sphere=False, env_rot_radian=env_rot_radian, fused=fused, **kw2)
        finally:
            self.min_near = saved_near
        ref_image = torch.cat([ref["image"], ref["weights_sum"][..., None]], -1)
        ref2ray = ref_mask[ray_mask]
        r_images = ref_image.new_zeros(1, ref2ray.shape[0], 4).masked_scatter(ref2ray[None, :, None], ref_image)
        kw3 = dict(kwargs, bg_color=0)
        res = self._run(rays_o[:, ray_mask, :], rays_d[:, ray_mask, :], get_normal_image=get_normal_image,
                        use_specular_color=use_specular_color, env_net_index=env_net_index, main_pass=True, r_images=r_images,
                        bg_sphere=False, env_rot_radian=env_rot_radian, fused=fused, **kw3)
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
