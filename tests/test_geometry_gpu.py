"""GPU parity of the geometry pipeline (envidr_geometry_eval / envidr_geometry_pass, through the C ABI) and of the
two-phase frames built on it (FusedRenderer.render_frame)."""
from pathlib import Path

import numpy as np
import pytest

from envidr_amd import scenes
from tests.util import rel_l2

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"
KEYS = ["image", "depth", "weights_sum", "normal_image", "diffuse_image", "specular_image", "roughness_image"]


@pytest.fixture(scope="module")
def scene():
    return scenes.toaster_scene()


@pytest.fixture(scope="module")
def renderer(scene):
    from envidr_amd.fused import FusedRenderer
    return FusedRenderer.from_scene(scene)


def _frame(renderer, rays_o, rays_d, env_rot=None, **kw):
    import torch
    res = renderer.render_frame(torch.from_numpy(rays_o).cuda(), torch.from_numpy(rays_d).cuda(), env_rot, **kw)
    torch.cuda.synchronize()
    out = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in res.items()}
    ws = out["weights_sum"][:, None]
    out["normal_image"] = out["normal_image"] * ws + (1 - ws)          # NeRFRenderer.render's final blend (renderer.py:529-530)
    out["roughness_image"] = out["roughness_image"][:, None]
    return out


@pytest.fixture(scope="module")
def renderer16(scene):
    """the same scene on the alternative 16-column geometry kernel (k_geo_eval16)"""
    from envidr_amd.fused import FusedOptions, FusedRenderer
    return FusedRenderer.from_scene(scene, FusedOptions(bound=scene.bound, grid_size=scene.grid_size, geometry_kernel="16"))


def test_sixteen_column_geometry_kernel_agrees(renderer, renderer16):
    """k_geo_eval16 against k_geo_eval32 on a frame: same integer trace, images within fp32 summation order (the few samples
    on a ReLU kink of the SDF network may flip: bounded like the other cross-implementation checks)"""
    rays_o, rays_d = scenes.camera_rays(96, 96, theta=70.0, phi=25.0)
    a, b = _frame(renderer, rays_o, rays_d, 0.2), _frame(renderer16, rays_o, rays_d, 0.2)
    assert np.array_equal(a["ray_cost"], b["ray_cost"]) and a["n_records"] == b["n_records"]
    for key in KEYS:
        err = rel_l2(b[key], a[key])
        assert err <= (5e-4 if key == "normal_image" else 2e-4), f"{key}: rel-L2 {err:.3e}"
    g = np.load(GOLD / "frame_toaster_48.npz")
    rays_o, rays_d = scenes.camera_rays(int(g["H"]), int(g["W"]), theta=float(g["theta"]), phi=float(g["phi"]))
    out = _frame(renderer16, rays_o, rays_d, None if np.isnan(g["env_rot"]) else float(g["env_rot"]))
    for key in KEYS:
        err = rel_l2(out[key], g[key].reshape(out[key].shape))
        assert err <= 1e-4, f"{key}: rel-L2 {err:.3e}"


def test_geometry_eval_matches_the_oracle_per_sample(scene, renderer):
    """envidr_geometry_eval (hash grid + SDF network forward / backward + density, normal, roughness) against the CPU
    oracle's per-sample chain (hash operator + torch fp32 layers + autograd normal) on points in and around the shell"""
    import torch
    from oracle.py import render_oracle as ro
    rng = np.random.default_rng(5)
    d = rng.normal(size=(4096, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    xyz = (d * rng.uniform(0.45, 0.8, size=(4096, 1))).astype(np.float32)
    xyz[:8] = [[1, 1, 1], [-1, -1, -1], [1, -1, 0.5], [0, 0, 0], [1.5, 0, 0], [0, -2, 0], [0.999999, 0.3, -0.2], [-1, 0.25, 1]]   # faces, corners, outside
    dirs = np.tile(np.array([[0, 0, 1]], np.float32), (4096, 1))
    want = ro.shade_samples(scene, xyz, dirs, ro.RenderOptions(ide_mode="exact"), None)
    dt = np.full(4096, 0.0034, np.float32)
    got = renderer.geometry_eval(torch.from_numpy(xyz).cuda(), torch.from_numpy(dt).cuda(), want=("alpha", "sigma", "normal", "roughness", "geo_feat"))
    torch.cuda.synchronize()
    sig = got["sigma"].cpu().numpy()
    assert rel_l2(sig, want["sigma"]) <= 1e-5
    assert rel_l2(got["roughness"].cpu().numpy(), want["roughness"].reshape(-1)) <= 1e-5
    inside = np.all(np.abs(xyz) <= 1, axis=1)
    n_err = np.abs(got["normal"].cpu().numpy() - want["normal"])[inside]
    assert np.mean(n_err) <= 1e-5 and np.quantile(n_err, 0.999) <= 1e-3          # unit normals: a tiny gradient amplifies rounding
    alpha = 1 - np.exp(-sig.astype(np.float64) * dt)
    assert np.allclose(got["alpha"].cpu().numpy(), alpha, rtol=2e-6, atol=1e-7)
    gn = np.linalg.norm(got["geo_feat"].cpu().numpy(), axis=1)[inside]          # (outside the cube the features are zero)
    assert np.all(np.isfinite(gn)) and np.abs(gn - 1).max() <= 1e-5, (np.abs(gn - 1).max(), int(np.argmax(np.abs(gn - 1))))


@pytest.mark.parametrize("tag", ["toaster_48", "toaster_rot_40"])
def test_frames_match_reference_frames(renderer, tag):
    """two-phase frames on the geometry pipeline against frames rendered by the reference itself: relative L2 <= 1e-4 on fp32
    RGB and on every auxiliary image (the north-star bound)"""
    g = np.load(GOLD / f"frame_{tag}.npz")
    H, W = int(g["H"]), int(g["W"])
    rays_o, rays_d = scenes.camera_rays(H, W, theta=float(g["theta"]), phi=float(g["phi"]))
    env_rot = None if np.isnan(g["env_rot"]) else float(g["env_rot"])
    out = _frame(renderer, rays_o, rays_d, env_rot)
    for key in KEYS:
        err = rel_l2(out[key], g[key].reshape(out[key].shape))
        assert err <= 1e-4, f"{key}: rel-L2 {err:.3e}"
    mse = float(np.mean((out["image"].astype(np.float64) - g["image"].reshape(-1, 3)) ** 2))
    assert -10 * np.log10(max(mse, 1e-30)) > 70.0


def test_integer_trace_is_the_oracles(scene, renderer):
    """The integer side of the path, pinned bit for bit: which samples a ray takes (occupancy decisions) and where it stops.
    Per-ray composited-sample counts and the (ray, index) set of the records equal the CPU oracle's run of the reference
    loop with one sample per iteration (raymarching.cu:839-944, :957-1046); weights agree to fp32 rounding."""
    import torch
    from oracle.py import render_oracle as ro
    rays_o, rays_d = scenes.camera_rays(36, 36, theta=75.0, phi=-10.0)
    want = ro.render_rays(scene, rays_o, rays_d, ro.RenderOptions(ide_mode="exact"), None, force_n_step=1)
    out = _frame(renderer, rays_o, rays_d)
    counts = out["ray_cost"].astype(np.int64)
    assert np.array_equal(counts, want["ray_counts"]), int(np.abs(counts - want["ray_counts"]).sum())
    assert out["n_records"] == int(want["ray_counts"].sum()) == want["n_samples"]
    st = renderer._frame
    M = out["n_records"]
    rec_ray, rec_idx = st["ray"][:M].cpu().numpy().astype(np.int64), st["idx"][:M].cpu().numpy().astype(np.int64)
    key = np.sort(rec_ray * 2048 + rec_idx)
    expect = np.concatenate([r * 2048 + np.arange(c) for r, c in enumerate(want["ray_counts"]) if c])
    assert np.array_equal(key, expect)                                  # every (ray, index) exactly once
    wsum = np.bincount(rec_ray, weights=st["w"][:M].cpu().numpy().astype(np.float64), minlength=rays_o.shape[0])
    assert np.allclose(wsum, want["weights_sum"], rtol=2e-5, atol=1e-6)
    for key_ in KEYS:
        err = rel_l2(out[key_], want[key_].reshape(out[key_].shape))
        assert err <= 2e-5, f"{key_}: rel-L2 {err:.3e}"


def test_full_size_counts_equal_the_persistent_kernel(renderer):
    """800x800: per-ray sample counts of the pipeline == those of envidr_render_rays (whose marcher is the bit-exact standalone
    operator's code) for every one of the 640 000 rays; images agree to fp32 rounding"""
    import torch
    ro_, rd_ = (torch.from_numpy(a).cuda() for a in scenes.camera_rays(800, 800))
    cost = torch.zeros(640000, dtype=torch.int16, device="cuda")
    one = {k: v.clone() for k, v in renderer.render(ro_, rd_, 0.3, extras=True, ray_cost=cost).items()}
    res = renderer.render_frame(ro_, rd_, 0.3)
    torch.cuda.synchronize()
    diff = (res["ray_cost"].int() - cost.int()).abs()
    # a ray stops when its transmittance falls below T_thresh: the two kernels round the densities differently (1e-7), so a
    # handful of rays may stop one sample apart; everything before that decision is integer-identical
    assert int((diff > 1).sum()) == 0 and int((diff == 1).sum()) <= 64, (int(diff.max()), int((diff > 0).sum()))
    assert int(res["n_records"]) == int(res["ray_cost"].int().sum())
    for key in ("image", "depth", "weights_sum", "normal_image", "diffuse_image", "specular_image", "roughness_image"):
        err = float((res[key] - one[key]).norm() / one[key].norm())
        # two fp32 evaluation orders of the hash interpolation: ~5e-6 of the samples sit within rounding of a ReLU kink of the
        # SDF network and get a different normal (and hence colours) from the two kernels
        assert err <= (5e-4 if key == "normal_image" else 2e-4 if key.endswith("_image") else 5e-5), f"{key}: {err:.2e}"


def test_frames_do_not_depend_on_the_hint(renderer):
    """the per-ray count hint only sizes allocations: exact, absent (16 samples, then chunks predicted per ray from its
    transmittance and last opacity), over-estimating (zero-filled slots inside the blocks' sample-major layout),
    under-estimating (extra rounds, predicted chunks again), all-ones and garbage hints all give the same frame, bit for bit"""
    import torch
    ro_, rd_ = (torch.from_numpy(a).cuda() for a in scenes.camera_rays(160, 160, theta=65.0, phi=35.0))
    N = ro_.shape[0]
    keys = ("image", "depth", "weights_sum", "normal_image", "diffuse_image", "specular_image", "roughness_image")
    want = {k: v.clone() for k, v in renderer.render_frame(ro_, rd_, 0.3, use_cost_hint=False).items() if k in keys or k == "ray_cost"}
    exact = want["ray_cost"].clone()
    st = renderer._frames[N]
    gen = torch.Generator(device="cuda").manual_seed(7)
    hints = {
        "exact": exact,
        "over": (exact.to(torch.int32) + torch.randint(0, 40, (N,), device="cuda", generator=gen, dtype=torch.int32)).to(exact.dtype),
        "under": (exact.to(torch.int32) // 2).to(exact.dtype),
        "garbage": torch.randint(0, 200, (N,), device="cuda", generator=gen, dtype=torch.int32).to(exact.dtype),
        "over on rays that miss": torch.full_like(exact, 25),
        "ones": torch.ones_like(exact),
    }
    for name, h in hints.items():
        st["cost"].copy_(h)
        for tag in list(st.get("costs", {})):
            st["costs"][tag].copy_(h)
        got = renderer.render_frame(ro_, rd_, 0.3, use_cost_hint=True)
        torch.cuda.synchronize()
        assert torch.equal(got["ray_cost"], exact), name
        for k in keys:
            assert torch.equal(got[k], want[k]), (name, k)


def test_rays_that_run_to_max_steps_are_complete_under_any_hint():
    """a ray whose sample count is max_steps (the cube's diagonal through a transparent, fully occupied grid) gets every one of
    its samples whatever its hint was: the last scheduled round hands out all that is left, not max_steps minus what an
    un-hinted ray would have had by then (a hint of 1 used to truncate such a ray by up to 15 samples)"""
    import dataclasses
    import torch
    from envidr_amd.fused import FusedOptions, FusedRenderer
    base = scenes.toaster_scene(sdf_bias=0.2)                     # sigma ~ 1e-7: nothing terminates on transmittance
    full = dataclasses.replace(base, bitfield=np.full_like(base.bitfield, 255))
    r = FusedRenderer.from_scene(full, FusedOptions(bound=full.bound, grid_size=full.grid_size, min_near=0.05))
    d0 = -np.ones(3) / np.sqrt(3)
    rng = np.random.default_rng(4)
    d = d0 + rng.normal(scale=2e-3, size=(96, 3)); d[:8] = d0
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ro_ = torch.tensor(np.tile([[1.5, 1.5, 1.5]], (96, 1)), dtype=torch.float32, device="cuda")
    rd_ = torch.tensor(d, dtype=torch.float32, device="cuda")
    want = {k: v.clone() for k, v in r.render_frame(ro_, rd_, None, use_cost_hint=False, samples_per_ray_hint=1100).items() if hasattr(v, "clone")}
    counts = want["ray_cost"].to(torch.int32)
    assert int(counts.max()) >= 1024 - 2 and int(counts.min()) > 900, (int(counts.max()), int(counts.min()))
    st = r._frames[96]
    for name, h in {"ones": torch.ones_like(want["ray_cost"]), "short": (counts // 3).to(torch.int16), "exact": want["ray_cost"].clone()}.items():
        st["cost"].copy_(h)
        for tag in list(st["costs"]):
            st["costs"][tag].copy_(h)
        got = r.render_frame(ro_, rd_, None, use_cost_hint=True, samples_per_ray_hint=1100)
        torch.cuda.synchronize()
        assert torch.equal(got["ray_cost"], want["ray_cost"]), (name, int((got["ray_cost"].int() - counts).abs().max()))
        for k in ("image", "depth", "weights_sum", "normal_image"):
            assert torch.equal(got[k], want[k]), (name, k)


def test_rays_are_independent_of_their_batch(renderer):
    """a ray's pixels do not depend on which other rays share its frame (blocks of 64 rays, sample-major slots, the per-ray
    hint): a sub-range rendered alone -- unaligned to the 64-ray blocks -- is bit-identical to the same rays inside the frame"""
    import torch
    ro_, rd_ = (torch.from_numpy(a).cuda() for a in scenes.camera_rays(400, 400, theta=80.0, phi=40.0))
    full = {k: v.clone() for k, v in renderer.render_frame(ro_, rd_, 0.4).items() if hasattr(v, "clone")}
    lo, hi = 30011, 101003
    part = renderer.render_frame(ro_[lo:hi].contiguous(), rd_[lo:hi].contiguous(), 0.4)
    torch.cuda.synchronize()
    for key in ("image", "depth", "weights_sum", "diffuse_image", "specular_image", "roughness_image", "ray_cost"):
        assert torch.equal(full[key][lo:hi], part[key]), key


@pytest.mark.parametrize("grid", [8, 16, 32])
def test_small_occupancy_grids(grid):
    """the one-cascade marcher + the occupied-box clipping on small grids (their bitfields are a few 16-byte loads): same
    per-ray counts as the general marcher of the single-kernel renderer, same images"""
    import dataclasses
    import torch
    from envidr_amd.fused import FusedOptions, FusedRenderer
    base = scenes.toaster_scene()
    small = dataclasses.replace(base, bitfield=scenes.occupancy_bitfield(scenes.shell(0.5, 0.12), H=grid), grid_size=grid)
    r = FusedRenderer.from_scene(small, FusedOptions(bound=small.bound, grid_size=grid))
    ro_, rd_ = (torch.from_numpy(a).cuda() for a in scenes.camera_rays(96, 96, theta=60.0, phi=15.0))
    cost = torch.zeros(ro_.shape[0], dtype=torch.int16, device="cuda")
    one = {k: v.clone() for k, v in r.render(ro_, rd_, 0.2, extras=True, ray_cost=cost).items()}
    res = r.render_frame(ro_, rd_, 0.2)
    torch.cuda.synchronize()
    assert int(res["ray_cost"].sum()) > 1000
    assert torch.equal(res["ray_cost"].to(torch.int32), cost.to(torch.int32))
    for key in ("image", "depth", "weights_sum"):
        assert rel_l2(res[key].cpu().numpy(), one[key].cpu().numpy()) <= 2e-4, key


def test_frame_that_does_not_fit_is_redone(renderer):
    import torch
    ro_, rd_ = (torch.from_numpy(a).cuda() for a in scenes.camera_rays(96, 96))
    want = {k: v.clone() for k, v in renderer.render_frame(ro_, rd_, None).items() if hasattr(v, "clone")}
    renderer.__dict__.pop("_frames", None); renderer.__dict__.pop("_frame_hints", None)
    got = renderer.render_frame(ro_, rd_, None, samples_per_ray_hint=0.5)           # far too small: grown and redone inside
    torch.cuda.synchronize()
    for key in ("image", "depth", "weights_sum", "normal_image"):
        assert torch.equal(got[key], want[key]), key
    assert renderer._frame["cap"] > 0.5 * 96 * 96
    renderer.__dict__.pop("_frames", None); renderer.__dict__.pop("_frame_hints", None)


def test_frame_edge_cases(renderer):
    import torch
    o = torch.tensor([[0.0, 0.0, -4.0]] * 70, device="cuda")
    d = torch.tensor([[0.0, 1.0, 0.0]] * 70, device="cuda")
    res = renderer.render_frame(o, d)                 # every ray misses the scene box
    torch.cuda.synchronize()
    assert torch.all(res["weights_sum"] == 0) and torch.all(res["image"] == 1.0) and res["n_records"] == 0
    ro_, rd_ = scenes.camera_rays(9, 7)
    a = renderer.render_frame(torch.from_numpy(ro_).cuda(), torch.from_numpy(rd_).cuda())
    img = a["image"].clone()
    one = renderer.render_frame(torch.from_numpy(ro_[31:32]).cuda(), torch.from_numpy(rd_[31:32]).cuda())
    torch.cuda.synchronize()
    assert torch.equal(one["image"][0], img[31])      # a ray's result does not depend on its batch
    g = renderer.render_frame(torch.from_numpy(ro_).cuda(), torch.from_numpy(rd_).cuda(), geometry_only=True)
    torch.cuda.synchronize()
    assert torch.equal(g["depth"], a["depth"]) and torch.equal(g["normal_image"], a["normal_image"])
    # a frame of ZERO rays (a tile shard of a rank that got no tile; a masked batch that came out empty) is an empty result, not an error
    z = renderer.render_frame(o[:0], d[:0])
    torch.cuda.synchronize()
    assert z["image"].shape == (0, 3) and z["depth"].shape == (0,) and z["n_records"] == 0
    again = renderer.render_frame(torch.from_numpy(ro_).cuda(), torch.from_numpy(rd_).cuda())
    torch.cuda.synchronize()
    assert torch.equal(again["image"], img)
    # the other frame schedules on the same degenerate inputs: zero rays, and a geometry cache that holds no sample
    assert renderer.render_two_phase(o[:0], d[:0])["image"].shape == (0, 3) and renderer.render(o[:0], d[:0])["image"].shape == (0, 3)
    cache = renderer.cache_geometry(o, d)                 # every ray misses
    lit = renderer.render_cached(cache, 0.3)
    torch.cuda.synchronize()
    assert int(cache.offsets[-1]) == 0 and torch.all(lit["image"] == 1.0) and torch.all(lit["weights_sum"] == 0)
    assert torch.equal(renderer.render_cached(renderer.cache_geometry(o[:0], d[:0]))["image"], torch.empty(0, 3, device="cuda"))


def test_no_environment_family_on_the_pipeline():
    """BASELINE configs[1] (SH view dir / normal into the specular head, no env MLP): geometry pipeline + heads-only record shading
    against the reference's frame and against the oracle on the kernel's schedule"""
    from envidr_amd.fused import FusedOptions, FusedRenderer
    from oracle.py import render_oracle as ro
    lego = scenes.lego_scene(seed=8)
    r = FusedRenderer.from_scene(lego, FusedOptions(dir_sh_degree=4))
    g = np.load(GOLD / "frame_lego_48.npz")
    H, W = int(g["H"]), int(g["W"])
    rays_o, rays_d = scenes.camera_rays(H, W, theta=float(g["theta"]), phi=float(g["phi"]))
    out = _frame(r, rays_o, rays_d)
    for key in KEYS:
        err = rel_l2(out[key], g[key].reshape(out[key].shape))
        assert err <= 1e-4, f"{key} vs reference frame: rel-L2 {err:.3e}"
    rays_o, rays_d = scenes.camera_rays(40, 40, theta=15.0, phi=-60.0)
    want = ro.render_rays(lego, rays_o, rays_d, ro.RenderOptions(), None, force_n_step=1)
    out = _frame(r, rays_o, rays_d)
    assert np.array_equal(out["ray_cost"].astype(np.int64), want["ray_counts"])
    for key in KEYS:
        err = rel_l2(out[key], want[key].reshape(out[key].shape))
        assert err <= 2e-5, f"{key} vs oracle: rel-L2 {err:.3e}"


def test_frame_buffers_converge_on_what_the_frames_use(renderer):
    """a first guess of the sample / record capacity that is far too large (here 100 per ray; 20 by default) is trimmed to 1.25x the peak the frames have needed once it
    has been more than twice that for 64 frames; the frames do not change, the per-ray hint survives the re-allocation"""
    import torch
    renderer.__dict__.pop("_frames", None); renderer.__dict__.pop("_frame_hints", None)
    ro_, rd_ = (torch.from_numpy(a).cuda() for a in scenes.camera_rays(128, 128, theta=20.0, phi=10.0))
    want = {k: v.clone() for k, v in renderer.render_frame(ro_, rd_, 0.2, samples_per_ray_hint=100.0).items() if hasattr(v, "clone")}
    cap0 = renderer._frame["cap"]
    for _ in range(70):
        got = renderer.render_frame(ro_, rd_, 0.2)
    torch.cuda.synchronize()
    st = renderer._frame
    assert st["cap"] < cap0 and st["cap"] >= st["peak"] and st["cap"] <= 2 * st["peak"] + 8192, (cap0, st["cap"], st["peak"])
    for k in ("image", "depth", "weights_sum", "normal_image", "ray_cost"):
        assert torch.equal(got[k], want[k]), k
    assert got["n_samples"] == got["n_records"]            # the hint was carried over: still one exact round
    renderer.__dict__.pop("_frames", None)


def test_image_width_hint_changes_the_block_layout_not_the_frame(renderer):
    """image_width: blocks of 64 rays become 8x8-pixel tiles (the samples a wave evaluates are then neighbours in both image
    directions); every output stays indexed by the ray's position in the list and keeps its bits -- hinted and cold, with a ray
    mask, and for sizes that do not qualify (ignored)"""
    import torch
    H, W = 96, 128
    ro_, rd_ = (torch.from_numpy(a).cuda() for a in scenes.camera_rays(H, W, theta=40.0, phi=15.0))
    keys = ("image", "depth", "weights_sum", "normal_image", "diffuse_image", "specular_image", "roughness_image", "ray_cost")
    for hint in (False, True):
        want = {k: v.clone() for k, v in renderer.render_frame(ro_, rd_, 0.3, use_cost_hint=hint).items() if k in keys}
        got = renderer.render_frame(ro_, rd_, 0.3, use_cost_hint=hint, image_width=W)
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(got[k], want[k]), (hint, k)
    mask = (torch.arange(H * W, device="cuda") % 3 != 0)
    want = {k: v.clone() for k, v in renderer.render_frame(ro_, rd_, 0.3, ray_mask=mask).items() if k in keys}
    got = renderer.render_frame(ro_, rd_, 0.3, ray_mask=mask, image_width=W)
    for k in keys:
        assert torch.equal(got[k], want[k]), k
    odd = renderer.render_frame(ro_[: 90 * W], rd_[: 90 * W], 0.3, image_width=W)         # 90 rows: not a multiple of 8 -> list order
    ref = renderer.render_frame(ro_[: 90 * W], rd_[: 90 * W], 0.3)
    assert torch.equal(odd["image"], ref["image"])


def test_two_cascades_and_growing_steps_on_the_pipeline():
    """the general marcher of the pipeline (MODE 0: two occupancy cascades; with dt_gamma > 0 the step grows along the ray) and render
    knobs away from the toaster defaults, against the oracle on the reference schedule with n_step = 1 -- and the same integer
    trace as the persistent kernel (per-ray sample counts)"""
    import torch
    from envidr_amd.fused import FusedOptions, FusedRenderer
    from oracle.py import render_oracle as ro
    rng = np.random.default_rng(12)
    offsets, pls = scenes.hash_level_offsets(desired_resolution=2 * 2048)
    base = scenes.toaster_scene(seed=12)
    scene2 = scenes.SceneParams(bitfield=scenes.occupancy_bitfield(scenes.shell(1.1, 0.12), bound=2.0, cascades=2), offsets=offsets,
                                per_level_scale=pls, table=rng.uniform(-0.1, 0.1, size=(int(offsets[-1]), 2)).astype(np.float32),
                                mlps=base.mlps, beta=0.02, bound=2.0, cascades=2)
    knobs = dict(bound=2.0, min_near=0.05, max_steps=384, dt_gamma=1 / 128, T_thresh=1e-3, enabled_levels=8, intensity_scale=0.7,
                 roughness_scale=1.5)
    r = FusedRenderer.from_scene(scene2, FusedOptions(**knobs))
    rays_o, rays_d = scenes.camera_rays(40, 40, theta=140.0, phi=-30.0, radius=4.0, scale=1.2)
    want = ro.render_rays(scene2, rays_o, rays_d, ro.RenderOptions(cascades=2, ide_mode="exact", **knobs), None, force_n_step=1)
    out = _frame(r, rays_o, rays_d)
    assert want["n_samples"] > 5000 and out["n_records"] == want["n_samples"]
    for key in KEYS:
        err = rel_l2(out[key], want[key].reshape(out[key].shape))
        assert err <= 2e-5, f"{key}: rel-L2 {err:.3e}"
    o, d = torch.from_numpy(rays_o).cuda(), torch.from_numpy(rays_d).cuda()
    cost = torch.zeros(rays_o.shape[0], dtype=torch.int16, device="cuda")
    r.render(o, d, None, ray_cost=cost)
    torch.cuda.synchronize()
    assert torch.equal(cost, r.render_frame(o, d)["ray_cost"])


def test_a_frame_far_denser_than_the_first_guess_is_regrown_until_it_fits():
    """a solid ball with T_thresh = 0: ~300 samples per ray hitting it against a first guess of 20 -- the buffers grow geometrically
    (the device only knows that a frame did not fit, not by how much) until the frame fits; found by tools/geo/fuzz_frames.py"""
    import torch
    from envidr_amd.fused import FusedOptions, FusedRenderer
    r = FusedRenderer.from_scene(scenes.lego_scene(shape=scenes.ball(), seed=5), FusedOptions(T_thresh=0.0, dir_sh_degree=4))
    ro_, rd_ = scenes.camera_rays(100, 100)
    ro, rd = torch.from_numpy(ro_).cuda(), torch.from_numpy(rd_).cuda()
    out = r.render_frame(ro, rd)
    torch.cuda.synchronize()
    hit = out["weights_sum"] > 0
    assert out["n_records"] > 100 * int(hit.sum()) and torch.isfinite(out["image"]).all()
    cost = torch.zeros(ro.shape[0], dtype=torch.int16, device="cuda")
    r.render(ro, rd, None, ray_cost=cost)
    torch.cuda.synchronize()
    assert torch.equal(cost, out["ray_cost"])


def test_random_batches_and_knobs_pipeline_against_the_persistent_kernel():
    """a short run of tools/geo/fuzz_frames.py (random ray batches: 1 .. 20 k rays, ragged tails, random cameras, model boxes, max_steps 1 .. 1024,
    T_thresh 0 .. 0.5, constant and growing steps, near planes, environment rotations, layout hints, random masks, garbage count hints):
    the two frame implementations of this library agree on the integer trace and -- outside the isolated ReLU-kink rays -- to 1e-4,
    masks are honoured, frames do not depend on hints (4 500 iterations of it: profiles/r03*/fuzz_frames.txt)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_frames", Path(__file__).resolve().parents[1] / "tools" / "geo" / "fuzz_frames.py")
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    bad, line = fuzz.run(80, 3)
    assert bad == 0, line


def test_zero_weight_records_are_not_shaded_and_frames_do_not_change():
    """On a sharp-density scene most records have a compositing weight of exactly 0 (alpha underflows in fp32): the pipeline
    gathers the others for shading (envidr_geometry_export.shade_list).  Every output is bit-identical to shading all records."""
    import torch
    from envidr_amd.fused import FusedOptions, FusedRenderer
    scene = scenes.toaster_scene(beta=1e-3, sdf_bias=0.065)
    ro, rd = (torch.from_numpy(a).cuda() for a in scenes.camera_rays(96, 96))
    frames = {}
    for skip in (True, False):
        r = FusedRenderer.from_scene(scene, FusedOptions(skip_zero_weight=skip))
        res = r.render_frame(ro, rd, 0.3, out={})
        torch.cuda.synchronize()
        frames[skip] = {k: v.clone() for k, v in res.items() if torch.is_tensor(v)}
        if skip:
            records, shaded = int(res["n_records"]), int(r._frame["shade_list"][0].item())
            w = r._frame["w"][:records]
            assert shaded == int((w != 0).sum().item())
            assert 0 < shaded < 0.8 * records, (shaded, records)          # a real share of the records is skipped on this scene
            listed = r._frame["shade_list"][1:1 + shaded].long()
            assert torch.equal(torch.sort(listed).values, torch.nonzero(w != 0).flatten())
    for k in ("image", "depth", "weights_sum", "normal_image", "diffuse_image", "specular_image", "roughness_image"):
        assert torch.equal(frames[True][k], frames[False][k]), k
    assert torch.isfinite(frames[True]["image"]).all()
