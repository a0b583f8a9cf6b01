"""envidr_amd/nerf/render_func/non_cuda_ray.py (the mirror of the reference's torch-only render function) against the REFERENCE's own
`non_cuda_ray.run` -- torch on the CPU on both sides, no kernel involved: the reference's function was executed on an analytic stand-in
model whose density and colour are functions of the sample position (tests/golden/make_golden.py golden_torch_only_resample ->
torch_only_resample.npz), with and without importance re-sampling, and on prescribed per-sample values (torch_only.npz, the `vr|` arrays);
here the mirror runs on the same stand-ins.  The one extension call inside, near_far_from_aabb, is answered with the fixture's (near, far)
on both sides."""
from pathlib import Path

import numpy as np
import pytest
import torch

GOLD = Path(__file__).parent / "golden"


class _Opt:
    debug = False
    backsdf_loss = False
    eikonal_loss = False


class _Shell:
    """the stand-in of make_golden.resample_stub: a soft shell of radius 1.2"""
    opt = _Opt()
    training = False
    aabb_train = aabb_infer = torch.tensor([-8.0, -8, -8, 8, 8, 8])
    min_near = 0.2
    density_scale = 1
    bg_radius = -1

    def density(self, xyzs, **kw):
        r = xyzs.norm(dim=-1, keepdim=True)
        return {"sigma": 60.0 * torch.exp(-((r - 1.2) / 0.12) ** 2), "normal": xyzs / r.clamp_min(1e-6)}

    def color(self, xyzs, dirs, mask=None, **kw):
        return 0.5 + 0.5 * torch.sin(3.0 * xyzs + dirs)


def _run_with(nears, fars, *args, **kw):
    from envidr_amd.nerf.render_func import non_cuda_ray

    class _NearFar:
        @staticmethod
        def near_far_from_aabb(rays_o, rays_d, aabb, min_near):
            return torch.from_numpy(nears.copy()), torch.from_numpy(fars.copy())

    saved = non_cuda_ray.raymarching
    non_cuda_ray.raymarching = _NearFar
    try:
        return non_cuda_ray.run(*args, **kw)
    finally:
        non_cuda_ray.raymarching = saved


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_mirror_of_the_torch_only_render_function_on_an_analytic_model(tag):
    g = np.load(GOLD / "torch_only_resample.npz")
    steps, up = (int(v) for v in g[f"{tag}|steps"])
    res = _run_with(g["nears"], g["fars"], _Shell(), torch.from_numpy(g["rays_o"]), torch.from_numpy(g["rays_d"]), num_steps=steps,
                    upsample_steps=up, bg_color=torch.from_numpy(g["bg"]), perturb=False, get_normal_image=True)
    for k in ("image", "depth", "weights_sum", "normal_image"):
        got, want = res[k].detach().numpy(), g[f"{tag}|{k}"]
        assert got.shape == want.shape, (k, got.shape, want.shape)
        # the same torch on the same CPU: differences are the order of a few elementwise operations
        assert np.abs(got - want).max() <= 2e-6, (tag, k, float(np.abs(got - want).max()))
    assert float(res["weights_sum"].max()) > 0.99 and float(res["weights_sum"].min()) < 1e-3      # rays through the shell and rays that miss it


def test_mirror_on_prescribed_samples():
    """the fixture the compositing operators are pinned on (torch_only.npz `vr|`): prescribed sigma / rgb / normal per sample, no re-sampling"""
    g = np.load(GOLD / "torch_only.npz")
    sigma, rgb, normal = g["vr|sigma"], g["vr|rgb"], g["vr|normal"]
    N, T = sigma.shape

    class _Table(_Shell):
        def density(self, xyzs, **kw):
            return {"sigma": torch.from_numpy(sigma).reshape(-1, 1), "normal": torch.from_numpy(normal).reshape(-1, 3)}

        def color(self, xyzs, dirs, mask=None, **kw):
            return torch.from_numpy(rgb).reshape(-1, 3)

    rng = np.random.default_rng(0)
    rays_o = rng.normal(size=(N, 3)).astype(np.float32)
    rays_d = rng.normal(size=(N, 3)).astype(np.float32)
    res = _run_with(g["vr|nears"], g["vr|fars"], _Table(), torch.from_numpy(rays_o), torch.from_numpy(rays_d), num_steps=T, upsample_steps=0,
                    bg_color=torch.from_numpy(g["vr|bg"]), perturb=False, get_normal_image=True)
    for k in ("image", "depth", "weights_sum", "normal_image"):
        assert np.abs(res[k].detach().numpy() - g[f"vr|{k}"]).max() <= 2e-6, k


def test_inverse_cdf_samples_against_numpy_interpolation():
    from envidr_amd.nerf.render_func.non_cuda_ray import inverse_cdf_samples
    rng = np.random.default_rng(2)
    edges = np.sort(rng.uniform(0.5, 3.0, size=(7, 33)).astype(np.float32), axis=1)
    w = rng.uniform(0, 1, size=(7, 32)).astype(np.float32) ** 4
    got = inverse_cdf_samples(torch.from_numpy(edges), torch.from_numpy(w), 24, True).numpy()
    m = w.astype(np.float64) + 1e-5
    m /= m.sum(1, keepdims=True)
    cdf = np.concatenate([np.zeros((7, 1)), np.cumsum(m, 1)], 1)
    u = np.linspace(0.5 / 24, 1 - 0.5 / 24, 24)
    want = np.stack([np.interp(u, cdf[i], edges[i].astype(np.float64)) for i in range(7)])
    assert np.abs(got - want).max() <= 2e-5 and np.all(np.diff(got, axis=1) >= 0)
    torch.manual_seed(0)
    rnd = inverse_cdf_samples(torch.from_numpy(edges), torch.from_numpy(w), 50, False).numpy()
    assert rnd.shape == (7, 50) and np.all(rnd >= edges[:, :1]) and np.all(rnd <= edges[:, -1:])
