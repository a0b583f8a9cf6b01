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

The training branch (`model.training`, reference cuda_ray.py:64-237) is `_run_cuda_train` below: march_rays_train ->
forward_sigma (normals through autograd with create_graph, i.e. the hash encoder's second-order backward) -> forward_color ->
composite_rays_train (custom backward), everything differentiable on the HIP operators.  The Trainer around it (losses,
optimisers, schedules) stays out of scope; what this branch returns is what the reference's Trainer consumes.
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


SQRT3 = 3 ** 0.5


def _run_cuda_train(self, rays_o, rays_d, prefix, dt_gamma, bg_color, perturb, force_all_rays, max_steps, T_thresh, use_specular_color,
                    early_stop_steps, ray_depth, main_pass, r_images, geometry_only, grad_ray, bg_sphere):
    """run_cuda, `if self.training` branch (reference nerf/render_func/cuda_ray.py:64-237, minus its debug plotting): returns
    image / depth / weights_sum plus the per-sample tensors the reference's losses read (sigmas, sdfs, sdf_gradients and, when
    a relative-sdf style loss is on, the relsdf block).  Loss switches are the attributes the reference's Trainer toggles on
    `opt` (eikonal_loss, backsdf_loss, relsdf_loss, orientation_loss, dist_bound); absent attributes count as off."""
    opt = self.opt
    flag = lambda name: bool(getattr(opt, name, False))
    N, device = rays_o.shape[0], rays_o.device
    use_relsdf_loss = flag("relsdf_loss") or flag("dist_bound")
    use_orientation_loss, use_backsdf_loss, use_eikonal_loss = flag("orientation_loss"), flag("backsdf_loss"), flag("eikonal_loss")
    use_neus = flag("use_neus_sdf")
    use_sdf_sigma_grad = (use_relsdf_loss or use_backsdf_loss or use_eikonal_loss or use_orientation_loss or self.use_normal_with_mlp
                          or self.use_n_dot_viewdir or self.use_reflected_dir or use_neus)
    if ray_depth is not None:
        dt = 2 * SQRT3 / max_steps
        valid_ray = ray_depth > 0
        nears = ((ray_depth - 4 * dt) * valid_ray).float().contiguous().view(-1)
        fars = ((ray_depth + 4 * dt) * valid_ray).float().contiguous().view(-1)
    else:
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_train, self.min_near)
    results = {}
    if self.bg_radius > 0 and bg_sphere:
        sph = raymarching.sph_from_ray(rays_o, rays_d, self.bg_radius)
        bg_color = self.background(sph, rays_d)
        results["sphere_bg"] = bg_color
    elif bg_color is None:
        bg_color = 1
    if not force_all_rays:
        counter = self.step_counter[self.local_step % 16]
        counter.zero_()
        self.local_step += 1
    else:
        counter = None
    with torch.no_grad():                                   # the marcher has no backward (reference :76)
        stratified = flag("stratified_sampling")
        xyzs, dirs, deltas, rays = raymarching.march_rays_train(rays_o, rays_d, self.bound, self.density_bitfield, self.cascade, self.grid_size,
                                                                nears, fars, counter, self.mean_count, (not stratified) and perturb, 128,
                                                                force_all_rays, dt_gamma, max_steps, early_stop_steps)
        if stratified:
            dt = 2 * SQRT3 / max_steps
            noise = (torch.rand_like(deltas[..., :1]) * 2 - 1) * 0.5 * dt
            shift = noise.roll(1, dims=-2) - noise
            deltas = deltas + shift
            xyzs = xyzs + shift * dirs
    if use_sdf_sigma_grad:
        xyzs.requires_grad = True
    scatter_idx = None
    if r_images is not None:
        scatter_idx = raymarching.get_scatter_idx(rays, rays.new_zeros(xyzs.shape[0])).long()
        r_images = r_images[0, scatter_idx]
    if grad_ray:
        if scatter_idx is None:
            scatter_idx = raymarching.get_scatter_idx(rays, rays.new_zeros(xyzs.shape[0])).long()
        dirs = rays_d[scatter_idx]
        o_s = rays_o[scatter_idx]
        scale = getattr(opt, "grad_rays_scale", 1.0)
        xyzs = xyzs - scale * o_s.detach() + scale * o_s
    sdfs, sigmas, geo_feats, normals, eikonal = self.forward_sigma(xyzs, use_sdf_sigma_grad=use_sdf_sigma_grad, dirs=dirs, dists=deltas[..., 0])
    sigmas = self.density_scale * sigmas
    roughness = getattr(self, "roughness", opt.default_roughness)
    weights = None
    if geometry_only:
        with torch.set_grad_enabled(not opt.detach_normal):
            weights_sum, depth, normal_image, weights = raymarching.composite_rays_train(sigmas, normals, deltas, rays, T_thresh, False, use_neus)
        depth = ((depth + nears) * (depth != 0)).view(*prefix)
        results["normal_image"] = F.normalize(normal_image, dim=-1).view(*prefix, 3)
        image = None
    else:
        n_enc, w_r_enc, n_dot, n_env_enc = self.get_color_mlp_extra_params(normals, dirs, roughness)
        rgbs = self.forward_color(geo_feats, dirs, n_enc, w_r_enc, n_dot, use_specular_color, n_env_enc=n_env_enc, r_images=r_images,
                                  roughness=roughness)
        ret_weights = use_backsdf_loss or use_relsdf_loss or use_orientation_loss or flag("weighted_eikonal")
        weights_sum, depth, image, weights = raymarching.composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh, ret_weights, use_neus)
        image = (image + (1 - weights_sum).unsqueeze(-1) * bg_color).view(*prefix, 3)
        depth = ((depth + nears) * (depth != 0)).view(*prefix)
    results.update(image=image, depth=depth, weights_sum=weights_sum.view(*prefix), sigmas=sigmas, sdfs=sdfs)
    if flag("cauchy_roughness_weighted"):
        results["roughness"] = roughness
    if use_eikonal_loss:
        results["sdf_gradients"] = eikonal * weights.detach()[..., None] if flag("weighted_eikonal") else eikonal
    if main_pass and not geometry_only and (use_relsdf_loss or use_backsdf_loss or use_orientation_loss):
        # consecutive-sample sdf differences against their first-order estimate along the ray (reference :168-212)
        M = sigmas.shape[0]
        point_mask = torch.ones_like(sigmas, dtype=torch.bool)
        ray_valid = (rays[:, 2] > 0) & (rays[:, 1] + rays[:, 2] < M)
        start = rays[ray_valid, 1]
        point_mask[(start + rays[ray_valid, 2] - 1).long()] = False            # a ray's last sample has no successor
        shifted = torch.roll(deltas, -1, 0)
        point_mask = point_mask & (shifted[:, 0] > 0) & (shifted[:, 1] > 0) & (shifted[:, 1] < 1.2 * shifted[:, 0])
        if self.obj_aabb is not None:
            point_mask = point_mask & (xyzs >= self.obj_aabb[:3]).all(-1) & (xyzs <= self.obj_aabb[3:]).all(-1)
        s_ = sdfs if self.use_sdf else -sigmas
        relsdf = torch.roll(s_, -1, dims=0) - s_
        cos = (dirs * normals).sum(-1)
        results.update(relsdf=relsdf[point_mask], est_relsdf=(shifted[:, 1] * cos.detach())[point_mask], cos=cos[point_mask],
                       sdf_weights=weights[point_mask], sdf_dist=shifted[point_mask, 1], sdfs=s_[point_mask])
    return results


def run_cuda(model, rays_o, rays_d, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False, max_steps=1024,
             T_thresh=1e-4, get_normal_image=False, use_specular_color=True, early_stop_steps=-1, ray_depth=None,
             main_pass=True, r_images=None, geometry_only=False, grad_ray=False, bg_sphere=True, env_rot_radian=None,
             fused=True, two_phase=None, ray_mask=None, frame_tag="", wait=True, frame_buffers="", reuse_geometry=None, **kwargs):
    self = model
    prefix = rays_o.shape[:-1]
    rays_o = rays_o.contiguous().view(-1, 3)
    rays_d = rays_d.contiguous().view(-1, 3)
    N, device = rays_o.shape[0], rays_o.device
    opt = self.opt
    if self.training:
        return _run_cuda_train(self, rays_o, rays_d, prefix, dt_gamma=dt_gamma, bg_color=bg_color, perturb=perturb, force_all_rays=force_all_rays,
                               max_steps=max_steps, T_thresh=T_thresh, use_specular_color=use_specular_color, early_stop_steps=early_stop_steps,
                               ray_depth=ray_depth, main_pass=main_pass, r_images=r_images, geometry_only=geometry_only, grad_ray=grad_ray,
                               bg_sphere=bg_sphere)
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
        # T_thresh / dt_gamma): plain descriptor fields, read by the kernels at launch -- and put back when the launches are
        # enqueued, so that direct users of model.fused_renderer() keep seeing opt's values, not the last call's
        saved_march = (fr.desc.max_steps, fr.desc.T_thresh, fr.desc.dt_gamma)
        fr.desc.max_steps, fr.desc.T_thresh, fr.desc.dt_gamma = int(max_steps), float(T_thresh), float(dt_gamma)
        fr.set_aabb(self.aabb_infer)               # the operator loop's near_far_from_aabb box, not just +-bound
        # The geometry pipeline (march rounds + sample-parallel hash / SDF kernel -> record shading -> composite;
        # FusedRenderer.render_frame) renders every fused configuration: both network families, the geometry-only first pass
        # and the reflected-radiance third pass of indirect rendering.  It keeps, per batch size, the per-ray sample counts of
        # the previous render as a sizing hint (video frames share their rays; outputs never depend on it).
        # `two_phase=False` asks for the single persistent kernel instead (envidr_render_rays).
        try:
            pipeline = two_phase is not False
            if pipeline:
                # `image_width` (not a reference argument; optional): the rays are a row-major image this wide -- a layout hint that
                # lets the pipeline form its blocks from 8x8-pixel tiles (same outputs)
                res = fr.render_frame(rays_o, rays_d, env_rot_radian, geometry_only=geometry_only,
                                      r_images=None if r_images is None else r_images[0], ray_mask=ray_mask, tag=frame_tag, wait=wait,
                                      image_width=int(kwargs.get("image_width", 0) or 0), buffers=frame_buffers, reuse_geometry=reuse_geometry)
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
        finally:
            fr.desc.max_steps, fr.desc.T_thresh, fr.desc.dt_gamma = saved_march
        image = res["image"] if tensor_bg is None else res["image"] + (1 - res["weights_sum"])[:, None] * tensor_bg
        out = {"image": image.view(*prefix, 3), "depth": res["depth"].view(*prefix), "weights_sum": res["weights_sum"].view(*prefix)}
        if sphere_bg is not None:
            out["sphere_bg"] = sphere_bg
        if geometry_only:
            out["image"] = None
            out["normal_image"] = res["normal_image"].view(*prefix, 3)
            out["_frame"] = res            # the raw frame: render_frame(reuse_geometry=...) shades its records again (indirect main pass)
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

    use_neus = bool(getattr(self.opt, "use_neus_sdf", False))

    def composite(st, n_alive, n_step, sigmas, colors, deltas, accum=True):
        raymarching.composite_rays(n_alive, n_step, st["alive"], st["t"], sigmas, colors, deltas, st["ws"], st["depth"], st["image"],
                                   T_thresh, use_neus, accum)         # NeuS: `sigmas` are section alphas (reference cuda_ray.py:308-340)

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
            self._normals_only = True          # everything below is detached: only d sdf / d xyz is taken (network.py forward_geometry)
            try:
                sdfs, sigmas, geo_feats, normals, _ = self.forward_sigma(xyzs, use_sdf_sigma_grad=True, dirs=dirs, dists=deltas[..., 0])
            finally:
                self._normals_only = False
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
