"""`run_cuda`, inference branch: the reference's march -> shade -> composite -> compact loop
(nerf/render_func/cuda_ray.py:238-359) in two forms.

  * fused (default when the model supports it): the geometry pipeline + record shading (envidr_amd.fused.FusedRenderer
    .render_frame -> envidr_geometry_pass / envidr_shade_records / envidr_composite_records), or ONE persistent-kernel
    launch for the whole ray batch (envidr_render_rays);
  * operator loop: the reference's loop structure on the HIP operators (`raymarching.march_rays`,
    model.forward_sigma / forward_color on HIP encoders + torch GEMMs, `raymarching.composite_rays`),
    with the boolean-mask compaction replaced by the device-side `compact_alive` (one 4-byte readback
    per iteration instead of a nonzero + gather + sync).  Used for model configurations the fused kernel
    does not implement (and on request, `fused=False`).

The training branch of run_cuda (march_rays_train + composite_rays_train + losses) belongs to the
Trainer and is out of scope; its operators exist in envidr_amd.raymarching.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from ... import raymarching


def _state(N, nears, device):
    ids = torch.arange(N, dtype=torch.int32, device=device)
    return {"ws": torch.zeros(N, device=device), "depth": torch.zeros(N, device=device), "image": torch.zeros(N, 3, device=device),
            "alive": ids, "buf": ids, "spare": torch.empty(N, dtype=torch.int32, device=device), "t": nears.clone()}


def fused_eligible(model, r_images=None, geometry_only=False, fused=True, ray_depth=None, perturb=False, bg_color=None, N=None,
                   max_steps=1024, **_) -> bool:
    """the preconditions under which run_cuda takes the fused path (one place: _render_indirect asks the same question
    before it commits to the ray-mask form of the three passes, which only the fused path implements)"""
    if not fused or ray_depth is not None or perturb or not 1 <= int(max_steps) <= 65535:
        return False
    if torch.is_tensor(bg_color) and not (bg_color.dim() == 2 and N is not None and bg_color.shape[0] == N):
        return False
    return bool(model.supports_fused(r_images=r_images, geometry_only=geometry_only))


def run_cuda(model, rays_o, rays_d, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False, max_steps=1024,
             T_thresh=1e-4, get_normal_image=False, use_specular_color=True, early_stop_steps=-1, ray_depth=None,
             main_pass=True, r_images=None, geometry_only=False, grad_ray=False, bg_sphere=True, env_rot_radian=None,
             fused=True, two_phase=None, ray_mask=None, frame_tag="", wait=True, **kwargs):
    self = model
    if self.training:
        raise NotImplementedError("run_cuda training branch is out of scope (operators are in envidr_amd.raymarching)")
    prefix = rays_o.shape[:-1]
    rays_o = rays_o.contiguous().view(-1, 3)
    rays_d = rays_d.contiguous().view(-1, 3)
    N, device = rays_o.shape[0], rays_o.device
    opt = self.opt
    sphere_bg = None
    if self.bg_radius > 0 and bg_sphere:
        # background model (reference cuda_ray.py:56-60): where the ray leaves the sphere of radius bg_radius -> 2-D grid
        # encoding + SH of the view direction -> bg MLP; blended with (1 - weights_sum) like any background colour
        with torch.no_grad():
            sph = raymarching.sph_from_ray(rays_o, rays_d, self.bg_radius)
            sphere_bg = self.background(sph, rays_d)
        bg_color = sphere_bg
    elif bg_color is None:
        bg_color = 1
    visual = list(opt.visual_items) if opt.use_diffuse else []

    # ---------------- fused persistent kernel ----------------
    # a per-ray background (the sphere model) is blended after the fused render: it runs with background 0
    tensor_bg = bg_color if (torch.is_tensor(bg_color) and bg_color.dim() == 2 and bg_color.shape[0] == N) else None
    if fused_eligible(self, r_images=r_images, geometry_only=geometry_only, fused=fused, ray_depth=ray_depth, perturb=perturb,
                      bg_color=bg_color, N=N, max_steps=max_steps):
        fr = self.fused_renderer()
        fr.desc.bg_color = 0.0 if tensor_bg is not None else float(bg_color)
        fr.desc.min_near = float(self.min_near)
        # per-call march parameters (the reflected pass of indirect rendering has its own max_steps; callers may pass their own
        # T_thresh / dt_gamma): plain descriptor fields, read by the kernels at launch
        fr.desc.max_steps, fr.desc.T_thresh, fr.desc.dt_gamma = int(max_steps), float(T_thresh), float(dt_gamma)
        fr.set_aabb(self.aabb_infer)               # the operator loop's near_far_from_aabb box, not just +-bound
        # The geometry pipeline (march rounds + sample-parallel hash / SDF kernel -> record shading -> composite;
        # FusedRenderer.render_frame) renders every fused configuration: both network families, the geometry-only first pass
        # and the reflected-radiance third pass of indirect rendering.  It keeps, per batch size, the per-ray sample counts of
        # the previous render as a sizing hint (video frames share their rays; outputs never depend on it).
        # `two_phase=False` asks for the single persistent kernel instead (envidr_render_rays).
        pipeline = two_phase is not False
        if pipeline:
            res = fr.render_frame(rays_o, rays_d, env_rot_radian, geometry_only=geometry_only,
                                  r_images=None if r_images is None else r_images[0], ray_mask=ray_mask, tag=frame_tag, wait=wait)
        else:
            if ray_mask is not None:
                raise NotImplementedError("ray_mask is a feature of the geometry pipeline (two_phase)")
            hints = self.__dict__.setdefault("_ray_cost_hints", {})
            key = (N, bool(geometry_only), r_images is not None)
            if key not in hints or hints[key].device != device:
                if len(hints) > 8:
                    hints.clear()
                hints[key] = torch.zeros(N, dtype=torch.int16, device=device)
            res = fr.render(rays_o, rays_d, env_rot_radian, extras=True, geometry_only=geometry_only,
                            r_images=None if r_images is None else r_images[0], ray_cost=hints[key])
        image = res["image"] if tensor_bg is None else res["image"] + (1 - res["weights_sum"])[:, None] * tensor_bg
        out = {"image": image.view(*prefix, 3), "depth": res["depth"].view(*prefix), "weights_sum": res["weights_sum"].view(*prefix)}
        if sphere_bg is not None:
            out["sphere_bg"] = sphere_bg
        if geometry_only:
            out["image"] = None
            out["normal_image"] = res["normal_image"].view(*prefix, 3)
            return out
        if get_normal_image:
            out["normal_image"] = res["normal_image"].view(*prefix, 3)
        if "diffuse" in visual:
            out["diffuse_image"] = res["diffuse_image"]
        if "specular" in visual:
            out["specular_image"] = res["specular_image"]
            out["roughness_image"] = res["roughness_image"][..., None]
        return out

    # ---------------- operator loop ----------------
    if ray_mask is not None:
        raise NotImplementedError("ray_mask is a feature of the fused geometry pipeline")
    nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_infer, self.min_near)
    main = _state(N, nears, device)
    extra = {}
    if get_normal_image and not geometry_only:
        extra["normal"] = _state(N, nears, device)
    if not geometry_only and "diffuse" in visual:
        extra["diffuse"] = _state(N, nears, device)
    if not geometry_only and "specular" in visual:
        extra["specular"] = _state(N, nears, device)
    count = torch.zeros(1, dtype=torch.int32, device=device)

    def compact(states):
        # device-side order-preserving compaction of every composited stream into its spare buffer (ping-pong).  All
        # streams composite the same densities and deltas, and a ray dies on its transmittance alone, so their alive lists
        # are identical: ONE survivor count (4 bytes) is read back per iteration to size the next one (the reference
        # re-derives each list with its own boolean mask + host sync, cuda_ray.py:326-341)
        for st in states:
            raymarching.compact_alive(st["alive"], st["spare"], count)
        n = int(count.item())
        for st in states:
            st["buf"], st["spare"] = st["spare"], st["buf"]
            st["alive"] = st["buf"][:n]

    def composite(st, n_alive, n_step, sigmas, colors, deltas, accum=True):
        raymarching.composite_rays(n_alive, n_step, st["alive"], st["t"], sigmas, colors, deltas, st["ws"], st["depth"], st["image"],
                                   T_thresh, False, accum)

    step = 0
    while step < max_steps:
        n_alive = main["alive"].shape[0]
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        align = 128
        _r = None
        if r_images is not None:
            _r = r_images[0, main["alive"].long()][:, None, :].expand(-1, n_step, -1).reshape(-1, r_images.shape[-1])
            align = -1
        xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, main["alive"], main["t"], rays_o, rays_d, self.bound,
                                                    self.density_bitfield, self.cascade, self.grid_size, nears, fars, align,
                                                    perturb if step == 0 else False, dt_gamma, max_steps)
        with torch.enable_grad():
            xyzs.requires_grad_(True)
            sdfs, sigmas, geo_feats, normals, _ = self.forward_sigma(xyzs, use_sdf_sigma_grad=True, dirs=dirs, dists=deltas[..., 0])
        roughness = self.roughness
        sigmas = (self.density_scale * sigmas).detach()
        normals = normals.detach()
        if geometry_only:
            composite(main, n_alive, n_step, sigmas, normals, deltas)
        else:
            with torch.no_grad():
                n_enc, w_r_enc, n_dot, n_env_enc = self.get_color_mlp_extra_params(normals, dirs, roughness, env_rot_radian)
                rgbs = self.forward_color(geo_feats.detach(), dirs, n_enc, w_r_enc, n_dot, use_specular_color,
                                          n_env_enc=n_env_enc, r_images=_r, roughness=roughness)
            composite(main, n_alive, n_step, sigmas, rgbs, deltas)
            if "diffuse" in extra:
                composite(extra["diffuse"], n_alive, n_step, sigmas, self.c_diffuse, deltas)
            if "specular" in extra:
                accum = True
                if torch.is_tensor(roughness):
                    deltas[..., 1:] = roughness.detach()       # roughness rides in the depth slot (reference :329-333)
                    accum = False
                composite(extra["specular"], n_alive, n_step, sigmas, self.c_specular, deltas, accum)
            if "normal" in extra:
                composite(extra["normal"], n_alive, n_step, sigmas, normals, deltas)
        compact([main, *extra.values()])
        step += n_step

    results = {"depth": main["depth"].view(*prefix), "weights_sum": main["ws"].view(*prefix)}
    if sphere_bg is not None:
        results["sphere_bg"] = sphere_bg
    if geometry_only:
        results["image"] = None
        results["normal_image"] = F.normalize(main["image"], dim=-1, eps=1e-10).view(*prefix, 3)
        return results
    results["image"] = (main["image"] + (1 - main["ws"]).unsqueeze(-1) * bg_color).view(*prefix, 3)
    if "normal" in extra:
        results["normal_image"] = F.normalize(extra["normal"]["image"], dim=-1, eps=1e-10).view(*prefix, 3)
    if "diffuse" in extra:
        results["diffuse_image"] = extra["diffuse"]["image"]
    if "specular" in extra:
        results["specular_image"] = extra["specular"]["image"]
        results["roughness_image"] = extra["specular"]["depth"][..., None]
    return results
