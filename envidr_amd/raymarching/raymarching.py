"""`raymarching` operator surface on the HIP library.

Same public names, argument orders, defaults and output-allocation rules as the reference's
raymarching/raymarching.py (functions exported at :49,80,104,126,155,164,246,310,367,394), so
`nerf/renderer.py` / `nerf/render_func/*` call sites work unchanged.  Differences, all deliberate:
inputs that are not on the GPU are moved there (the reference does `.cuda()`), and a missing
extension raises instead of falling back.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from .. import _lib

__all__ = ["near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "get_scatter_idx",
           "march_rays_train", "composite_rays_train", "march_rays", "composite_rays", "compact_alive"]


def _gpu(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_cuda else t.cuda()


def _rays(rays_o, rays_d):
    return _gpu(rays_o).float().contiguous().view(-1, 3), _gpu(rays_d).float().contiguous().view(-1, 3)


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    """rays [N,3] x aabb [6] -> nears, fars [N] (FLT_MAX where the box is missed)"""
    rays_o, rays_d = _rays(rays_o, rays_d)
    N = rays_o.shape[0]
    nears = torch.empty(N, dtype=torch.float32, device=rays_o.device)
    fars = torch.empty(N, dtype=torch.float32, device=rays_o.device)
    _lib.call("near_far_from_aabb", rays_o, rays_d, _gpu(aabb).float().contiguous(), N, min_near, nears, fars)
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius):
    """far intersection with the background sphere -> (theta, phi) in [-1, 1]^2, [N,2]"""
    rays_o, rays_d = _rays(rays_o, rays_d)
    N = rays_o.shape[0]
    coords = torch.empty(N, 2, dtype=torch.float32, device=rays_o.device)
    _lib.call("sph_from_ray", rays_o, rays_d, radius, N, coords)
    return coords


def morton3D(coords):
    coords = _gpu(coords).int().contiguous()
    N = coords.shape[0]
    indices = torch.empty(N, dtype=torch.int32, device=coords.device)
    _lib.call("morton3D", coords, N, indices)
    return indices


def morton3D_invert(indices):
    indices = _gpu(indices).int().contiguous()
    N = indices.shape[0]
    coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
    _lib.call("morton3D_invert", indices, N, coords)
    return coords


def packbits(grid, thresh, bitfield=None):
    """density grid [C, H^3] -> occupancy bitfield uint8 [C * H^3 / 8]"""
    grid = _gpu(grid).float().contiguous()
    N = grid.shape[0] * grid.shape[1] // 8
    if bitfield is None:
        bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
    _lib.call("packbits", grid, N, thresh, bitfield)
    return bitfield


def get_scatter_idx(rays, source):
    _lib.call("get_scatter_idx", rays, rays.shape[0], source)
    return source


def march_rays_train(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                     perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024, early_stop_steps=-1):
    """training marcher: returns xyzs [M,3], dirs [M,3], deltas [M,2], rays [N,3] = (id, offset, count)"""
    rays_o, rays_d = _rays(rays_o, rays_d)
    density_bitfield = _gpu(density_bitfield).contiguous()
    dev = rays_o.device
    N = rays_o.shape[0]
    M = N * max_steps
    if not force_all_rays and mean_count > 0:
        if align > 0:
            mean_count += align - mean_count % align
        M = mean_count
    xyzs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    dirs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    deltas = torch.zeros(M, 2, dtype=torch.float32, device=dev)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    if step_counter is None:
        step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
    noises = torch.rand(N, dtype=torch.float32, device=dev) if perturb else torch.zeros(N, dtype=torch.float32, device=dev)
    early_stop_steps = max_steps if early_stop_steps <= 0 else early_stop_steps
    _lib.call("march_rays_train", rays_o, rays_d, density_bitfield, bound, dt_gamma, max_steps, early_stop_steps, N, C, H, M,
              nears, fars, xyzs, dirs, deltas, rays, step_counter, noises)
    if force_all_rays or mean_count <= 0:
        m = int(step_counter[0].item())
        if align > 0:
            m += align - m % align
        xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
    return xyzs, dirs, deltas, rays


class _CompositeRaysTrain(Function):
    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4, ret_weights=False, input_alpha=False, accum_deltas=True):
        sigmas, rgbs = sigmas.float().contiguous(), rgbs.float().contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        accum_deltas, input_alpha = int(accum_deltas), int(input_alpha)
        dev = sigmas.device
        weights_sum = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        weights = torch.zeros_like(sigmas) if ret_weights else torch.zeros(0, device=dev)
        _lib.call("composite_rays_train_forward", sigmas, rgbs, deltas, rays, M, N, T_thresh, accum_deltas, input_alpha,
                  weights_sum, depth, image, weights if ret_weights else None)
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        ctx.meta = (M, N, T_thresh, accum_deltas, input_alpha)
        return weights_sum, depth, image, weights

    @staticmethod
    def backward(ctx, grad_weights_sum, grad_depth, grad_image, grad_weights):
        sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N, T_thresh, accum_deltas, input_alpha = ctx.meta
        grad_sigmas = torch.zeros_like(sigmas)
        grad_rgbs = torch.zeros_like(rgbs)
        _lib.call("composite_rays_train_backward", grad_weights_sum.contiguous(), grad_image.contiguous(), grad_depth.contiguous(),
                  sigmas, rgbs, deltas, rays, weights_sum, image, depth, M, N, T_thresh, grad_sigmas, grad_rgbs, accum_deltas,
                  input_alpha)
        return grad_sigmas, grad_rgbs, None, None, None, None, None, None


composite_rays_train = _CompositeRaysTrain.apply


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1,
               perturb=False, dt_gamma=0, max_steps=1024):
    """inference marcher: up to n_step samples for each of the first n_alive ids; outputs are
    zero-filled and padded with the reference's rule `M += align - M % align`."""
    rays_o, rays_d = _rays(rays_o, rays_d)
    dev = rays_o.device
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    xyzs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    dirs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    deltas = torch.zeros(M, 2, dtype=torch.float32, device=dev)
    noises = torch.rand(n_alive, dtype=torch.float32, device=dev) if perturb else torch.zeros(n_alive, dtype=torch.float32, device=dev)
    _lib.call("march_rays", n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H,
              density_bitfield, near, far, xyzs, dirs, deltas, noises)
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2,
                   input_alpha=False, accum_deltas=True):
    """in-place compositing of one march_rays batch; writes -1 into rays_alive for finished rays"""
    _lib.call("composite_rays", n_alive, n_step, T_thresh, int(accum_deltas), int(input_alpha), rays_alive, rays_t,
              sigmas.float().contiguous(), rgbs.float().contiguous(), deltas, weights_sum, depth, image)
    return tuple()


def compact_alive(rays_alive: torch.Tensor, out: torch.Tensor | None = None, count: torch.Tensor | None = None):
    """device-side `rays_alive[rays_alive >= 0]` (order preserving, no host sync):
    returns (buffer, count tensor); the first count[0] entries of buffer are the survivors."""
    n = rays_alive.shape[0]
    out = torch.empty_like(rays_alive) if out is None else out
    count = torch.zeros(1, dtype=torch.int32, device=rays_alive.device) if count is None else count
    _lib.call("compact_alive", n, rays_alive, out, count)
    return out, count
