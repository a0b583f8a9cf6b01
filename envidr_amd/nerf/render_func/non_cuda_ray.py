"""`run`: the reference's render function for `cuda_ray = False` (nerf/render_func/non_cuda_ray.py:13-181, dispatched at
nerf/renderer.py:368-371) -- no occupancy grid: `num_steps` uniform samples between the ray's box entry and exit, optionally
`upsample_steps` more drawn from the piecewise-constant distribution the first pass's compositing weights define (inverse-CDF
sampling, nerf/render_func/utils.py:4-40), all of them composited with the cumprod recurrence.  Encoders and the box intersection are
this package's HIP operators; the rest is torch, as in the reference.

What it feeds the colour network is the reference's: geometry feature, view direction and the RAW normal -- no reflected direction, no n.v,
no encoded normal -- so it renders configurations whose colour network expects nothing else (tests/golden/plain_like.ini); for
toaster.ini-style networks the reference's function fails on the colour network's input width, and so does this one.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from ... import raymarching


def inverse_cdf_samples(edges: torch.Tensor, weights: torch.Tensor, count: int, deterministic: bool) -> torch.Tensor:
    """`count` positions per row from the piecewise-linear CDF over `edges` [N, E] with bin masses `weights` [N, E - 1] (+ 1e-5 each): at
    the mid-points of `count` equal probability slices when `deterministic`, else at uniform random probabilities (utils.py sample_pdf)."""
    mass = weights + 1e-5
    mass = mass / mass.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(mass[..., :1]), torch.cumsum(mass, -1)], -1)                     # [N, E]
    if deterministic:
        u = torch.linspace(0.5 / count, 1.0 - 0.5 / count, steps=count, device=cdf.device).expand(*cdf.shape[:-1], count)
    else:
        u = torch.rand(*cdf.shape[:-1], count, device=cdf.device)
    u = u.contiguous()
    upper = torch.searchsorted(cdf, u, right=True)
    lo = (upper - 1).clamp(min=0)
    hi = upper.clamp(max=cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    e_lo, e_hi = torch.gather(edges, -1, lo), torch.gather(edges, -1, hi)
    span = c_hi - c_lo
    span = torch.where(span < 1e-5, torch.ones_like(span), span)
    return e_lo + (u - c_lo) / span * (e_hi - e_lo)


def _compositing_weights(z: torch.Tensor, sigma: torch.Tensor, last_step: torch.Tensor, density_scale: float) -> torch.Tensor:
    """w_i = alpha_i prod_{j<i} (1 - alpha_j + 1e-15), alpha = 1 - exp(-delta sigma), delta_i = z_{i+1} - z_i (the last: `last_step`)"""
    delta = torch.cat([z[..., 1:] - z[..., :-1], last_step * torch.ones_like(z[..., :1])], -1)
    alpha = 1 - torch.exp(-delta * density_scale * sigma)
    through = torch.cumprod(torch.cat([torch.ones_like(alpha[..., :1]), 1 - alpha + 1e-15], -1), -1)[..., :-1]
    return alpha * through


def run(model, rays_o, rays_d, num_steps=128, upsample_steps=128, bg_color=None, perturb=False, get_normal_image=False,
        use_specular_color=True, **kwargs):
    """rays_o, rays_d [B, N, 3] (B == 1 in the reference).  Returns {image [B,N,3], depth [B,N], weights_sum [N], normal_image [N,3] or None}
    (+ sdf_gradients when training with the eikonal loss), like the reference's function."""
    self = model
    opt = self.opt
    prefix = rays_o.shape[:-1]
    rays_o = rays_o.contiguous().view(-1, 3)
    rays_d = rays_d.contiguous().view(-1, 3)
    N, device = rays_o.shape[0], rays_o.device
    # (the reference sets use_neus_density = True here, which makes the gradient pass unconditional)
    want_grad = True
    aabb = self.aabb_train if self.training else self.aabb_infer
    nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)
    nears, fars = nears.unsqueeze(-1), fars.unsqueeze(-1)
    z = nears + (fars - nears) * torch.linspace(0.0, 1.0, num_steps, device=device).unsqueeze(0)               # [N, T]
    sample_dist = (fars - nears) / num_steps
    if perturb:
        z = z + (torch.rand(z.shape, device=device) - 0.5) * sample_dist

    def points(zv):
        p = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * zv.unsqueeze(-1)
        p = torch.min(torch.max(p, aabb[:3]), aabb[3:])                            # kept inside the box
        if get_normal_image or want_grad:
            p.requires_grad_(True)
        return p

    def query(p, zv, steps, grad):
        # (directions repeated ray-major against sample-major points and `dists` = the sample depths, as the reference passes them;
        #  only the NeuS density reads either)
        out = self.density(p.reshape(-1, 3), use_sdf_sigma_grad=grad, dirs=rays_d.repeat(steps, 1), dists=zv.reshape(-1))
        return {k: (None if v is None else v.view(N, steps, -1)) for k, v in out.items()}

    xyzs = points(z)
    fields = query(xyzs, z, num_steps, want_grad)
    if upsample_steps > 0:
        with torch.no_grad():
            w = _compositing_weights(z, fields["sigma"].squeeze(-1), sample_dist, self.density_scale)
            mids = z[..., :-1] + 0.5 * (z[..., 1:] - z[..., :-1])
            new_z = inverse_cdf_samples(mids, w[:, 1:-1], upsample_steps, deterministic=not self.training).detach()
        new_xyzs = points(new_z)
        new_fields = query(new_xyzs, new_z, upsample_steps, True)
        z, order = torch.sort(torch.cat([z, new_z], 1), dim=1)
        pick = lambda a, b: torch.gather(torch.cat([a, b], 1), 1, order.unsqueeze(-1).expand(-1, -1, a.shape[-1]))
        xyzs = pick(xyzs, new_xyzs)
        fields = {k: (None if v is None else pick(v, new_fields[k])) for k, v in fields.items()}

    with torch.set_grad_enabled(self.training):
        weights = _compositing_weights(z, fields["sigma"].squeeze(-1), sample_dist, self.density_scale)            # [N, S]
        dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
        flat = {k: (None if v is None else v.reshape(-1, v.shape[-1])) for k, v in fields.items()}
        rgbs = self.color(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), mask=(weights > 1e-4).reshape(-1), **flat).view(N, -1, 3)
        weights_sum = weights.sum(-1)
    normal_image = None
    if get_normal_image:
        normal_image = F.normalize(torch.sum(weights.unsqueeze(-1) * flat["normal"].reshape(*weights.shape, 3), dim=-2), dim=-1, eps=1e-10)
    depth = torch.sum(weights * ((z - nears) / (fars - nears)).clamp(0, 1), dim=-1)
    image = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)
    if self.bg_radius > 0:
        bg_color = self.background(raymarching.sph_from_ray(rays_o, rays_d, self.bg_radius), rays_d)
    elif bg_color is None:
        bg_color = 1
    image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
    out = {"depth": depth.view(*prefix), "image": image.view(*prefix, 3), "weights_sum": weights_sum, "normal_image": normal_image}
    if opt.eikonal_loss and self.training:
        out["sdf_gradients"] = flat["sdf_gradients"]
    return out
