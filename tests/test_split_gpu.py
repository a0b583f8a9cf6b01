"""GPU parity of the optional split-precision shading mode (FusedOptions.env_precision = "f16x2": environment MLP on the fp16
matrix cores with (hi, lo) operand pairs, csrc/mlp_split.hip.h + shade_split.hip).  Same bar as the fp32 path: relative L2
<= 1e-4 against the reference's own shading chains and frames."""
from pathlib import Path

import numpy as np
import pytest

from envidr_amd import scenes
from tests.util import rel_l2

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"
KEYS = ["image", "depth", "weights_sum", "normal_image", "diffuse_image", "specular_image", "roughness_image"]


@pytest.fixture(scope="module")
def renderer():
    from envidr_amd.fused import FusedRenderer
    return FusedRenderer.from_scene(scenes.toaster_scene())


@pytest.mark.parametrize("tag", ["shading_toaster", "shading_toaster_rot"])
def test_split_shading_matches_reference_chains(renderer, tag):
    """the reference's per-sample colours (1 536 samples of its own shading chain) from geometry inputs, both precisions"""
    import torch
    g = np.load(GOLD / f"{tag}.npz")
    env_rot = None if np.isnan(g["env_rot"]) else float(g["env_rot"])
    args = [torch.from_numpy(np.ascontiguousarray(g[k])).cuda() for k in ("normal", "dirs", "geo_feat")]
    rough = torch.from_numpy(np.ascontiguousarray(g["roughness"]).reshape(-1)).cuda()
    got = {p: renderer.shade(*args, rough, env_rot, env_precision=p) for p in ("fp32", "f16x2")}
    torch.cuda.synchronize()
    for p, res in got.items():
        for k in ("c_diffuse", "c_specular"):
            err = rel_l2(res[k].cpu().numpy(), g[k].reshape(-1, 3))
            assert err <= 1e-4, f"{p} {k}: rel-L2 {err:.3e}"
    for k in ("c_diffuse", "c_specular"):       # the two precisions against each other: far inside the bound
        a, b = got["fp32"][k].cpu().numpy(), got["f16x2"][k].cpu().numpy()
        assert np.isfinite(b).all() and rel_l2(b, a) <= 2e-5, (k, rel_l2(b, a))


def test_split_shading_on_ragged_sizes(renderer):
    """sample counts that do not fill a workgroup / wave, shared geometry feature and roughness (stride 0), M = 1"""
    import torch
    rng = np.random.default_rng(5)
    for M in (1, 31, 33, 127, 129, 1000):
        n = rng.normal(size=(M, 3)).astype(np.float32); n /= np.linalg.norm(n, axis=1, keepdims=True)
        d = rng.normal(size=(M, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
        geo = rng.normal(size=12).astype(np.float32); geo /= np.linalg.norm(geo)
        t = [torch.from_numpy(x).cuda() for x in (n, d, geo)]
        a = renderer.shade(*t, 0.25, 0.4, env_precision="fp32")
        b = renderer.shade(*t, 0.25, 0.4, env_precision="f16x2")
        torch.cuda.synchronize()
        for k in ("c_diffuse", "c_specular"):
            assert rel_l2(b[k].cpu().numpy(), a[k].cpu().numpy()) <= 2e-5, (M, k)


@pytest.mark.parametrize("tag", ["toaster_48", "toaster_rot_40"])
def test_split_frames_match_reference_frames(renderer, tag):
    import torch
    g = np.load(GOLD / f"frame_{tag}.npz")
    H, W = int(g["H"]), int(g["W"])
    rays_o, rays_d = scenes.camera_rays(H, W, theta=float(g["theta"]), phi=float(g["phi"]))
    env_rot = None if np.isnan(g["env_rot"]) else float(g["env_rot"])
    res = renderer.render_frame(torch.from_numpy(rays_o).cuda(), torch.from_numpy(rays_d).cuda(), env_rot, env_precision="f16x2")
    torch.cuda.synchronize()
    out = {k: res[k].cpu().numpy() for k in ("image", "diffuse_image", "specular_image")}
    for key, val in out.items():
        err = rel_l2(val, g[key].reshape(val.shape))
        assert err <= 1e-4, f"{key}: rel-L2 {err:.3e}"
    mse = float(np.mean((out["image"].astype(np.float64) - g["image"].reshape(-1, 3)) ** 2))
    assert -10 * np.log10(max(mse, 1e-30)) > 70.0


def _random_shading_inputs(M, seed):
    import torch
    rng = np.random.default_rng(seed)
    n = rng.normal(size=(M, 3)).astype(np.float32); n /= np.linalg.norm(n, axis=1, keepdims=True)
    d = rng.normal(size=(M, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    geo = rng.normal(size=(M, 12)).astype(np.float32); geo /= np.linalg.norm(geo, axis=1, keepdims=True)
    rough = rng.uniform(0.0, 1.0, size=M).astype(np.float32)
    return [torch.from_numpy(x).cuda() for x in (n, d, geo)], torch.from_numpy(rough).cuda()


@pytest.mark.parametrize("hidden_env,ide_deg", [(256, 5), (160, 4)])
def test_two_group_kernel_has_the_bits_of_the_one_group_kernel(hidden_env, ide_deg):
    """"f16x2" (csrc/shade_split2.hip: layers fused in pairs, eight waves of 256 registers on one weight stream) against "f16x2_v1" (csrc/shade_split.hip:
    one group at a time): the same (hi, lo) split, the same three products per step in the same order into every accumulator -- the colours must
    be IDENTICAL, on sample counts that leave waves / workgroups / the last round ragged and on many rounds per workgroup"""
    import torch
    from envidr_amd.fused import FusedOptions, FusedRenderer
    r = FusedRenderer.from_scene(scenes.toaster_scene(hidden_env=hidden_env, ide_deg=ide_deg), FusedOptions(ide_degree=ide_deg))
    for M, rot in ((1, None), (33, 0.3), (127, None), (129, 1.1), (1000, 0.4), (256 * 128 + 77, 2.0), (700_001, 0.7)):
        args, rough = _random_shading_inputs(M, M)
        a = r.shade(*args, rough, rot, env_precision="f16x2")
        b = r.shade(*args, rough, rot, env_precision="f16x2_v1")
        c = r.shade(*args, rough, rot, env_precision="fp32")
        torch.cuda.synchronize()
        for k in ("c_diffuse", "c_specular"):
            assert torch.isfinite(a[k]).all()
            assert torch.equal(a[k], b[k]), (M, k, float((a[k] - b[k]).abs().max()))
            assert rel_l2(a[k].cpu().numpy(), c[k].cpu().numpy()) <= 2e-6, (M, k)


def test_two_group_kernel_is_deterministic_and_frames_agree():
    """the same frame three times through the fused-pair kernel (bit-identical: no atomics, no schedule dependence), and equal to the one-group
    kernel's frame; a frame with zero-weight skipping (the list of records to shade, which only the fused-pair form honours) equals the frame
    that shades every record"""
    import torch
    from envidr_amd.fused import FusedOptions, FusedRenderer
    rays_o, rays_d = (torch.from_numpy(x).cuda() for x in scenes.camera_rays(96, 96))
    r = FusedRenderer.from_scene(scenes.toaster_scene())
    frames = [r.render_frame(rays_o, rays_d, 0.2, out={}, env_precision="f16x2")["image"].clone() for _ in range(3)]
    old = r.render_frame(rays_o, rays_d, 0.2, out={}, env_precision="f16x2_v1")["image"].clone()
    f32 = r.render_frame(rays_o, rays_d, 0.2, out={}, env_precision="fp32")["image"].clone()
    assert torch.equal(frames[0], frames[1]) and torch.equal(frames[0], frames[2]) and torch.equal(frames[0], old)
    assert rel_l2(frames[0].cpu().numpy(), f32.cpu().numpy()) <= 1e-6
    sharp = scenes.toaster_scene(beta=1e-3, sdf_bias=0.065)
    img = {}
    for skip in (True, False):
        fr = FusedRenderer.from_scene(sharp, FusedOptions(skip_zero_weight=skip))
        img[skip] = fr.render_frame(rays_o, rays_d, 0.2, out={}, env_precision="f16x2")["image"].clone()
        if skip:
            listed, records = int(fr._frame["shade_list"][0].item()), int(fr._frame["last"][1])
            assert 0 < listed < records
    assert torch.equal(img[True], img[False])
