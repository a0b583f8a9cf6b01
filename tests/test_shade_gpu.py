"""envidr_shade_samples: the shading half of the loop on known geometry.
  * BASELINE configs[0]: the reference's demo.ipynb cell 17 (surface rendering of the unit sphere with the shipped
    demo/ weights), golden image produced by executing the notebook's own code cells on the CPU;
  * per-sample parity with the reference's forward_color on the toaster goldens."""
from pathlib import Path

import numpy as np
import pytest

from envidr_amd import scenes
from tests.util import rel_l2

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def sequential(g, name):
    """layers of an nn.Sequential(Linear, ReLU, ..., Linear) state_dict stored in the fixture"""
    idx = sorted({int(k.split("/")[1].split(".")[0]) for k in g.files if k.startswith(name + "/")})
    return [(g[f"{name}/{i}.weight"], g[f"{name}/{i}.bias"]) for i in idx]


def test_demo_sphere_surface_rendering():
    """BASELINE configs[0] through the PRODUCT entry point (envidr_amd.nerf.render_func.sph_ray.render_surface: sphere hit ->
    FusedShader -> composition), against the notebook's own images"""
    import torch
    from envidr_amd.fused import FusedShader
    from envidr_amd.nerf.render_func import sph_ray
    g = np.load(GOLD / "demo_sphere.npz")
    res = int(g["res"])
    pose = scenes.nerf_matrix_to_ngp(scenes.pose_spherical(float(g["theta"]), -float(g["phi"]), float(g["radius"])), scale=1.0)
    ro, rd = scenes.get_rays(pose, scenes.intrinsics_for(res, res), res, res)
    ro, rd = torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda()
    geo_feat, kappa_inv = sph_ray.material_features(sequential(g, "sdf_net"), torch.from_numpy(g["xyz_encoding"]).cuda(), float(g["roughness"]),
                                                    float(g["metallic"]), [float(v) for v in g["base_color"]])
    assert abs(float(kappa_inv) - float(g["kappa_inv"])) < 1e-6
    shader = FusedShader({"env": sequential(g, "env_net"), "diffuse": sequential(g, "diffuse_net"),
                          "specular": sequential(g, "specular_net")}, ide_degree=4, diffuse_kappa_inv=0.64)
    out = sph_ray.render_surface(shader, ro, rd, geo_feat, kappa_inv)
    torch.cuda.synchronize()
    assert np.array_equal(out["mask"].cpu().numpy(), g["mask"])
    for key, name in [("diffuse", "diffuse_image"), ("specular", "specular_image"), ("image", "image")]:
        err = rel_l2(out[name].cpu().numpy(), g[key])
        assert err <= 1e-4, f"{key}: rel-L2 {err:.3e}"
    # SURVEY.md section 6 anchor (400 x 400): recomputed by the generator from the same notebook run
    assert np.allclose(g["mean_rgb_400"], [0.62849, 0.70200, 0.82242], atol=2e-5)
    # rays that all miss: background only
    miss = sph_ray.render_surface(shader, ro[:5] * 0 + torch.tensor([0.0, 0.0, 4.0], device="cuda"), ro[:5] * 0 + torch.tensor([0.0, 1.0, 0.0], device="cuda"),
                                  geo_feat, kappa_inv)
    assert not miss["mask"].any() and torch.all(miss["image"] == 1)


@pytest.mark.parametrize("tag", ["toaster", "toaster_rot"])
def test_shade_matches_reference_forward_color(tag):
    """per-sample c_diffuse / c_specular from the reference's normals, geo_feat and roughness (goldens)"""
    import torch
    from envidr_amd.fused import FusedRenderer
    g = np.load(GOLD / f"shading_{tag}.npz")
    r = FusedRenderer.from_scene(scenes.toaster_scene())
    env_rot = None if np.isnan(g["env_rot"]) else float(g["env_rot"])
    cuda = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out = r.shade(cuda(g["normal"]), cuda(g["dirs"]), cuda(g["geo_feat"]), cuda(g["roughness"].reshape(-1)), env_rot)
    torch.cuda.synchronize()
    assert rel_l2(out["c_diffuse"].cpu().numpy(), g["c_diffuse"]) <= 1e-5
    # reflected-direction IDE at tiny roughness: the reference's own fp32 cancellation noise (DESIGN.md "IDE numerics")
    assert rel_l2(out["c_specular"].cpu().numpy(), g["c_specular"]) <= 1e-4
    # ragged size and the empty call
    out2 = r.shade(cuda(g["normal"][:77]), cuda(g["dirs"][:77]), cuda(g["geo_feat"][:77]), cuda(g["roughness"].reshape(-1)[:77]), env_rot)
    torch.cuda.synchronize()
    assert torch.equal(out2["c_diffuse"], out["c_diffuse"][:77]) and torch.equal(out2["c_specular"], out["c_specular"][:77])
    out3 = r.shade(cuda(g["normal"][:0]), cuda(g["dirs"][:0]), cuda(g["geo_feat"][:0]), cuda(g["roughness"].reshape(-1)[:0]), env_rot)
    assert out3["c_diffuse"].shape == (0, 3)


def test_geometry_cache_relights_bit_exactly():
    """march + hash + SDF once per camera (geometry-only render that exports its samples), then shading
    + compositing per environment rotation; every rotation must reproduce render() of the same rays bit for bit"""
    import torch
    from envidr_amd.fused import FusedRenderer
    r = FusedRenderer.from_scene(scenes.toaster_scene())
    ro, rd = (torch.from_numpy(a).cuda() for a in scenes.camera_rays(96, 96, theta=50.0, phi=-25.0))
    cache = r.cache_geometry(ro, rd, samples_per_ray_hint=2.0)            # too small on purpose: exercises the retry
    direct = r.render(ro, rd, None, extras=True, stats=True)
    torch.cuda.synchronize()
    composited = int(cache.offsets[-1])
    assert composited == cache.n_samples and 0 < composited <= int(direct["stats"][0])   # tail mode may shade a few more
    assert torch.all(cache.offsets[1:] >= cache.offsets[:-1])
    for rot in (None, 0.7, 3.9):
        want = {k: v.clone() for k, v in r.render(ro, rd, rot, extras=True).items()}
        got = r.render_cached(cache, rot)
        torch.cuda.synchronize()
        for k in ("image", "diffuse_image", "specular_image", "depth", "weights_sum", "normal_image"):
            assert torch.equal(got[k], want[k]), f"rot {rot}: {k}"
    # the cache is environment-independent: swapping the environment changes the cached render like the direct one
    assert not torch.equal(r.render_cached(cache, 0.7)["image"], r.render_cached(cache, 3.9)["image"].clone())


BUILT_ENV_SHAPES = [(5, 256), (4, 160), (5, 128), (4, 128)]      # (IDE degree, hidden width) of launch_shade / envidr_env_mlp_forward


@pytest.mark.parametrize("ide_deg,hidden", BUILT_ENV_SHAPES)
def test_every_built_environment_shape_shades_like_the_oracle(ide_deg, hidden):
    """each (IDE degree, hidden width) instantiation of the shading kernel against the oracle's forward_color chain on seeded weights:
    the hand-over form of the environment pass (even tile counts) only applies when the first layer is long enough for it --
    (4, 128) once ran it with 3 step-major steps and never loaded three of the next layer's bias tiles"""
    import torch
    from envidr_amd.fused import FusedShader
    from oracle.py import render_oracle as ro
    scene = scenes.toaster_scene(hidden_env=hidden, ide_deg=ide_deg, seed=11)
    mlps = {k: scene.mlps[k] for k in ("env", "diffuse", "specular")}
    shader = FusedShader(mlps, ide_degree=ide_deg, diffuse_kappa_inv=0.64)
    rng = np.random.default_rng(5)
    unit = lambda v: v / np.linalg.norm(v, axis=1, keepdims=True)
    M = 777
    n, d, gf = unit(rng.normal(size=(M, 3))), unit(rng.normal(size=(M, 3))), unit(rng.normal(size=(M, 12)))
    rough = rng.uniform(0.02, 1, M)
    cuda = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    for rot in (None, 0.7):
        want = ro.shade_surface(mlps, n, d, gf, rough.reshape(-1, 1), ro.RenderOptions(ide_mode="exact", ide_deg=ide_deg), rot)
        got = shader.shade(cuda(n), cuda(d), cuda(gf), cuda(rough), rot)
        torch.cuda.synchronize()
        for key in ("c_diffuse", "c_specular"):
            assert np.abs(got[key].cpu().numpy().astype(np.float64) - want[key]).max() <= 2e-5, (key, rot)


@pytest.mark.parametrize("ide_deg,hidden", BUILT_ENV_SHAPES)
def test_env_mlp_operator_equals_the_torch_layers(ide_deg, hidden):
    """envidr_env_mlp_forward (the shading kernels' environment pass as an operator) against torch's Linear / ReLU chain, ragged and
    unaligned sizes, the weight cache following in-place parameter updates, and the dispatch rule of the network mirror"""
    import torch
    import torch.nn as nn
    from envidr_amd import fused
    from envidr_amd.nerf import network
    torch.manual_seed(3)
    k = (2 ** ide_deg - 1 + ide_deg) * 2
    assert (k, hidden) in fused.ENV_MLP_SHAPES
    net = nn.ModuleList([nn.Linear(k, hidden), nn.Linear(hidden, hidden), nn.Linear(hidden, hidden), nn.Linear(hidden, 12)]).cuda()

    def chain(x):
        for i, lin in enumerate(net):
            x = lin(x)
            if i != 3:
                x = torch.relu(x)
        return x
    assert fused.env_mlp_supported(net)
    with torch.no_grad():
        for M in (1, 63, 64, 65, 4099, 50001):
            x = torch.randn(M + 1, k, device="cuda")[1:]          # a view that is not 16-byte aligned when k * 4 is not (k = 38)
            y, want = fused.env_mlp_forward(net, x), chain(x)
            assert y.shape == (M, 12) and float((y - want).norm() / want.norm()) <= 2e-6
        x = torch.randn(5000, k, device="cuda")
        before = fused.env_mlp_forward(net, x)
        net[1].weight.mul_(0.5)                                    # an in-place update: the cached blob must follow
        after = fused.env_mlp_forward(net, x)
        assert not torch.equal(before, after) and float((after - chain(x)).norm() / chain(x).norm()) <= 2e-6
        assert fused.env_mlp_forward(net, x[:0]).shape == (0, 12)
        # the mirror's dispatch: operator without autograd on large batches, torch layers otherwise (same values to fp32 rounding)
        assert torch.equal(network._run_mlp(net, x), after)
    small = network._run_mlp(net, x[:100].clone().requires_grad_(True))
    assert small.requires_grad and float((small.detach() - after[:100]).norm() / after[:100].norm()) <= 2e-6
    assert not fused.env_mlp_supported(nn.ModuleList([nn.Linear(k, 64), nn.Linear(64, 64), nn.Linear(64, 64), nn.Linear(64, 12)]).cuda())
