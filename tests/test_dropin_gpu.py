"""The reference-shaped Python surface (envidr_amd.nerf.*, envidr_amd.raymarching, encoders) on the GPU:
`NeRFNetwork.render()` -- fused kernel and operator loop -- against frames rendered by the reference."""
from pathlib import Path

import numpy as np
import pytest

from envidr_amd import scenes
from tests.util import rel_l2

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"
KEYS = ["image", "depth", "weights_sum", "normal_image", "diffuse_image", "specular_image", "roughness_image"]


def build_model(scene, **opt_overrides):
    import torch
    from envidr_amd.nerf.network import NeRFNetwork
    from envidr_amd.nerf.options import toaster_options
    opt = toaster_options(**opt_overrides)
    m = NeRFNetwork(encoding="hashgrid", encoding_dir=opt.encoding_dir, bound=opt.bound, cuda_ray=opt.cuda_ray, density_scale=1,
                    min_near=opt.min_near, density_thresh=opt.density_thresh, bg_radius=opt.bg_radius, use_sdf=opt.use_sdf,
                    hidden_dim=opt.hidden_dim, num_layers=opt.num_layers, num_layers_color=opt.num_layers_color,
                    hidden_dim_color=opt.hidden_dim_color, num_levels=opt.num_levels, geo_feat_dim=opt.geo_feat_dim, opt=opt)
    # load the scene exactly like a reference checkpoint would be loaded: through state_dict keys
    sd = {"encoder.embeddings": torch.from_numpy(scene.table), "sdf_density.beta": torch.tensor(scene.beta),
          "density_bitfield": torch.from_numpy(scene.bitfield)}
    for name, attr in [("sdf", "sdf_net"), ("env", "env_net"), ("diffuse", "diffuse_net"), ("specular", "color_net"), ("renv", "renv_net")]:
        for i, (W, b) in enumerate(scene.mlps.get(name, [])):
            sd[f"{attr}.{i}.weight"] = torch.from_numpy(W)
            sd[f"{attr}.{i}.bias"] = torch.from_numpy(b)
    if opt.use_neus_sdf:
        del sd["sdf_density.beta"]                 # the NeuS density has `variance` instead (set from init_variance)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected
    return m.cuda().eval(), opt


@pytest.fixture(scope="module")
def model_and_opt():
    return build_model(scenes.toaster_scene())


def test_two_phase_schedule_through_the_drop_in_surface(model_and_opt):
    """run_cuda picks the two-phase schedule for large batches; forced here on a small frame: same frame as the reference's"""
    import torch
    model, opt = model_and_opt
    g = np.load(GOLD / "frame_toaster_rot_40.npz")
    H, W = int(g["H"]), int(g["W"])
    ro, rd = scenes.camera_rays(H, W, theta=float(g["theta"]), phi=float(g["phi"]))
    res = model.render(torch.from_numpy(ro).cuda()[None], torch.from_numpy(rd).cuda()[None], staged=True, bg_color=1, perturb=False,
                       get_normal_image=True, env_rot_radian=float(g["env_rot"]), two_phase=True, max_steps=opt.max_steps,
                       T_thresh=opt.T_thresh, dt_gamma=opt.dt_gamma)
    torch.cuda.synchronize()
    for key in KEYS:
        err = rel_l2(res[key].detach().cpu().numpy().reshape(H * W, -1), g[key].reshape(H * W, -1))
        assert err <= 1e-4, f"{key}: rel-L2 {err:.3e}"


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("tag", ["toaster_48", "toaster_rot_40"])
def test_render_matches_reference_frames(model_and_opt, tag, fused):
    import torch
    model, opt = model_and_opt
    g = np.load(GOLD / f"frame_{tag}.npz")
    H, W = int(g["H"]), int(g["W"])
    ro, rd = scenes.camera_rays(H, W, theta=float(g["theta"]), phi=float(g["phi"]))
    env_rot = None if np.isnan(g["env_rot"]) else float(g["env_rot"])
    res = model.render(torch.from_numpy(ro).cuda()[None], torch.from_numpy(rd).cuda()[None], staged=True, bg_color=1, perturb=False,
                       get_normal_image=True, env_rot_radian=env_rot, fused=fused, max_steps=opt.max_steps, T_thresh=opt.T_thresh,
                       dt_gamma=opt.dt_gamma)
    torch.cuda.synchronize()
    assert res["image"].shape == (1, H * W, 3) and res["depth"].shape == (1, H * W)
    for key in KEYS:
        got = res[key].detach().cpu().numpy().reshape(H * W, -1)
        want = g[key].reshape(H * W, -1)
        err = rel_l2(got, want)
        assert err <= 1e-4, f"{key} (fused={fused}): rel-L2 {err:.3e}"


def test_per_sample_chain_matches_reference(model_and_opt):
    """forward_sigma / get_color_mlp_extra_params / forward_color on HIP encoders + rocBLAS GEMMs"""
    import torch
    model, _ = model_and_opt
    g = np.load(GOLD / "shading_toaster.npz")
    x = torch.from_numpy(g["xyz"]).cuda().requires_grad_(True)
    d = torch.from_numpy(g["dirs"]).cuda()
    sdfs, sigmas, geo, normals, _ = model.forward_sigma(x, use_sdf_sigma_grad=True, dirs=d)
    n_enc, w_r_enc, n_dot, n_env_enc = model.get_color_mlp_extra_params(normals, d, model.roughness, None)
    rgb = model.forward_color(geo, d, n_enc, w_r_enc, n_dot, True, n_env_enc=n_env_enc, roughness=model.roughness)
    got = {"sdf": sdfs, "sigma": sigmas, "geo_feat": geo, "normal": normals, "roughness": model.roughness, "c_diffuse": model.c_diffuse,
           "c_specular": model.c_specular, "rgb": rgb, "n_env_enc": n_env_enc}
    for k, v in got.items():
        want = g[k].reshape(tuple(v.shape))
        assert rel_l2(v.detach().cpu().numpy(), want) <= 1e-4, k
    # the reflected-direction IDE at near-zero roughness carries the reference's own fp32 cancellation
    # noise in its l = 16 terms (DESIGN.md "IDE numerics"): compare it where that noise is attenuated
    rough = g["roughness"].reshape(-1)
    sel = rough > 0.03
    assert rel_l2(w_r_enc.detach().cpu().numpy()[sel], g["w_r_enc"][sel]) <= 1e-3


def test_instance_norm_feature_activations_match_reference():
    """geo_feat_act = env_feat_act = instanceNorm (reference network.py:436-440, 542-546, 601-605): not a configuration the fused
    kernels are built for, so render() takes the operator loop -- per-sample chain and a frame against the reference's"""
    import torch
    model, opt = build_model(scenes.toaster_scene(seed=2), geo_feat_act="instanceNorm", env_feat_act="instanceNorm")
    assert not model.supports_fused()
    g = np.load(GOLD / "shading_toaster_inorm.npz")
    x = torch.from_numpy(g["xyz"]).cuda().requires_grad_(True)
    d = torch.from_numpy(g["dirs"]).cuda()
    sdfs, sigmas, geo, normals, _ = model.forward_sigma(x, use_sdf_sigma_grad=True, dirs=d)
    n_enc, w_r_enc, n_dot, n_env_enc = model.get_color_mlp_extra_params(normals, d, model.roughness, None)
    rgb = model.forward_color(geo, d, n_enc, w_r_enc, n_dot, True, n_env_enc=n_env_enc, roughness=model.roughness)
    for k, v in {"sdf": sdfs, "geo_feat": geo, "normal": normals, "roughness": model.roughness, "c_diffuse": model.c_diffuse,
                 "c_specular": model.c_specular, "rgb": rgb}.items():
        assert rel_l2(v.detach().cpu().numpy(), g[k].reshape(tuple(v.shape))) <= 1e-4, k
    assert abs(float(geo.detach().mean(-1).abs().max())) < 1e-5                        # instance-normalised: zero mean per sample
    f = np.load(GOLD / "frame_toaster_inorm_32.npz")
    H, W = int(f["H"]), int(f["W"])
    ro, rd = scenes.camera_rays(H, W, theta=float(f["theta"]), phi=float(f["phi"]))
    res = model.render(torch.from_numpy(ro).cuda()[None], torch.from_numpy(rd).cuda()[None], staged=True, bg_color=1, perturb=False,
                       get_normal_image=True, max_steps=opt.max_steps, T_thresh=opt.T_thresh, dt_gamma=opt.dt_gamma)
    torch.cuda.synchronize()
    for key in KEYS:
        err = rel_l2(res[key].detach().cpu().numpy().reshape(H * W, -1), f[key].reshape(H * W, -1))
        assert err <= 1e-4, f"{key}: rel-L2 {err:.3e}"


def test_neus_geometric_init_and_skip_layer_variants_match_reference():
    """forward_geometry's other forms (reference network.py:46-102 NeuS section alphas, 153-222 geometric_init = weight-normalised layers +
    Softplus(100), 417 skip_layers) -- none selected by a shipped config, all three at once here: the per-sample chain and a frame through
    run_cuda's operator loop (the compositors take input_alpha) against the reference's own; loaded through weight_g / weight_v keys"""
    import torch
    g = np.load(GOLD / "shading_variants.npz")
    scene = scenes.toaster_scene(seed=9)
    scene.mlps["sdf"] = []                                                       # (installed below under the weight-norm keys)
    model, opt = build_model(scene, use_neus_sdf=True, geometric_init=True, skip_layers=[1], init_variance=float(g["init_variance"]))
    assert not model.supports_fused() and [tuple(l.weight.shape) for l in model.sdf_net] == [(32, 32), (64, 64), (15, 64)]
    sd = {}
    for i in range(3):
        W = torch.from_numpy(g[f"sdf/{i}.weight"])
        sd[f"sdf_net.{i}.weight_v"], sd[f"sdf_net.{i}.weight_g"], sd[f"sdf_net.{i}.bias"] = W, W.norm(dim=1, keepdim=True), torch.from_numpy(g[f"sdf/{i}.bias"])
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected
    model = model.cuda()
    x = torch.from_numpy(g["xyz"]).cuda().requires_grad_(True)
    d = torch.from_numpy(g["dirs"]).cuda()
    sdfs, alphas, geo, normals, _ = model.forward_sigma(x, use_sdf_sigma_grad=True, dirs=d, dists=torch.from_numpy(g["dists"]).cuda())
    n_enc, w_r_enc, n_dot, n_env_enc = model.get_color_mlp_extra_params(normals, d, model.roughness, None)
    rgb = model.forward_color(geo, d, n_enc, w_r_enc, n_dot, True, n_env_enc=n_env_enc, roughness=model.roughness)
    for k, v in {"sdf": sdfs, "alpha": alphas, "geo_feat": geo, "normal": normals, "roughness": model.roughness, "rgb": rgb}.items():
        assert rel_l2(v.detach().cpu().numpy(), g[k].reshape(tuple(v.shape))) <= 1e-4, (k, rel_l2(v.detach().cpu().numpy(), g[k].reshape(tuple(v.shape))))
    f = np.load(GOLD / "frame_variants_32.npz")
    H, W = int(f["H"]), int(f["W"])
    ro, rd = scenes.camera_rays(H, W, theta=float(f["theta"]), phi=float(f["phi"]))
    res = model.render(torch.from_numpy(ro).cuda()[None], torch.from_numpy(rd).cuda()[None], staged=True, bg_color=1, perturb=False,
                       get_normal_image=True, max_steps=opt.max_steps, T_thresh=opt.T_thresh, dt_gamma=opt.dt_gamma)
    torch.cuda.synchronize()
    for key in KEYS:
        err = rel_l2(res[key].detach().cpu().numpy().reshape(H * W, -1), f[key].reshape(H * W, -1))
        assert err <= 1e-4, f"{key}: rel-L2 {err:.3e}"


def test_encoder_modules_have_reference_surface():
    import torch
    from envidr_amd.encoding import get_encoder
    x = torch.rand(100, 3, device="cuda")
    for name, kw, dim in [("hashgrid_diff", {}, 32), ("hashgrid", {}, 32), ("tiledgrid", {"num_levels": 4}, 8),
                          ("frequency", {"multires": 6}, 39), ("sphere_harmonics", {"degree": 4}, 16),
                          ("integrated_dir", {"degree": 5}, 72), ("frequency", {"multires": 0}, 3)]:
        enc, d = get_encoder(name, **kw)
        assert d == dim, name
        if isinstance(enc, torch.nn.Module):
            enc = enc.cuda()
        y = enc(torch.nn.functional.normalize(x, dim=-1) if name in ("sphere_harmonics", "integrated_dir") else x)
        assert y.shape == (100, dim) and torch.isfinite(y).all(), name
    # gradients flow through the grid encoders (first and second order for the hash encoder)
    enc, _ = get_encoder("hashgrid_diff", num_levels=4, log2_hashmap_size=12)
    enc = enc.cuda()
    enc.embeddings.data.uniform_(-1, 1)
    xi = (torch.rand(50, 3, device="cuda") * 1.8 - 0.9).requires_grad_(True)
    y = enc(xi).sum()
    gx, = torch.autograd.grad(y, xi, create_graph=True)
    (gx ** 2).sum().backward()
    # like the reference, the double backward reaches the table (and the upstream gradient), not the inputs
    assert torch.isfinite(enc.embeddings.grad).all() and enc.embeddings.grad.abs().sum() > 0


def test_tightened_marching_box_is_honoured_by_the_fused_paths():
    """aabb_infer smaller than [-bound, bound]^3 (opt.marching_aabb / checkpoints): the fused paths intersect rays with the
    SAME box as the operator loop's near_far_from_aabb (different near / far, fewer samples), not with the bound cube"""
    import torch
    model, opt = build_model(scenes.toaster_scene(seed=9))
    model.aabb_infer = torch.tensor([-0.45, -0.6, -0.5, 0.55, 0.4, 0.65], dtype=torch.float32, device="cuda")
    ro_, rd_ = scenes.camera_rays(40, 40, theta=50.0, phi=-25.0)
    ro, rd = torch.from_numpy(ro_).cuda()[None], torch.from_numpy(rd_).cuda()[None]
    kw = dict(staged=True, bg_color=1, perturb=False, get_normal_image=True, max_steps=opt.max_steps, T_thresh=opt.T_thresh, dt_gamma=opt.dt_gamma)
    loop = model.render(ro, rd, fused=False, **kw)
    full = build_model(scenes.toaster_scene(seed=9))[0].render(ro, rd, fused=False, **kw)
    assert rel_l2(loop["image"].cpu().numpy(), full["image"].cpu().numpy()) > 1e-2          # the box matters on this view
    for two_phase in (None, False):            # geometry pipeline, single persistent kernel
        got = model.render(ro, rd, fused=True, two_phase=two_phase, **kw)
        torch.cuda.synchronize()
        for key in ("image", "depth", "weights_sum"):
            err = rel_l2(got[key].cpu().numpy().reshape(1600, -1), loop[key].cpu().numpy().reshape(1600, -1))
            assert err <= 3e-5, f"two_phase={two_phase} {key}: {err:.2e}"


@pytest.mark.parametrize("fused", [True, False])
def test_background_sphere_branch_matches_reference(fused):
    """bg_radius > 0: sph_from_ray + 2-D grid_encode + sh_encode -> background MLP, blended with (1 - weights_sum); against the
    reference's own render of the same model (cuda_ray.py:56-62, network.py:343-367,727-742)"""
    import torch
    g = np.load(GOLD / "frame_toaster_bg_40.npz")
    model, opt = build_model(scenes.toaster_scene(seed=6), bg_radius=float(g["bg_radius"]))
    assert model.bg_net is not None and tuple(model.encoder_bg.embeddings.shape) == g["bg_table"].shape
    with torch.no_grad():
        model.encoder_bg.embeddings.data = torch.from_numpy(g["bg_table"]).cuda()
        for i, lin in enumerate(model.bg_net):
            lin.weight.data = torch.from_numpy(g[f"bg_w{i}"]).cuda()
    H, W = int(g["H"]), int(g["W"])
    ro_, rd_ = scenes.camera_rays(H, W, theta=float(g["theta"]), phi=float(g["phi"]))
    res = model.render(torch.from_numpy(ro_).cuda()[None], torch.from_numpy(rd_).cuda()[None], staged=True, bg_color=None, perturb=False,
                       get_normal_image=True, fused=fused, max_steps=opt.max_steps, T_thresh=opt.T_thresh, dt_gamma=opt.dt_gamma)
    torch.cuda.synchronize()
    assert model.supports_fused()
    assert np.abs(res["sphere_bg"].cpu().numpy() - g["sphere_bg"]).max() <= 1e-5
    for key in ("image", "depth", "weights_sum"):
        err = rel_l2(res[key].cpu().numpy().reshape(H * W, -1), g[key].reshape(H * W, -1))
        assert err <= 1e-4, f"{key}: rel-L2 {err:.3e}"


def test_hash_encoder_first_order_backward_reaches_the_table():
    """`enc(x).sum().backward()` (no double backward) must fill embeddings.grad: the table gradient of a weighted sum of the
    outputs equals the oracle's hash_encode_backward scatter (hashencoder.cu:257-343), and a plain training step would otherwise
    silently never update the table"""
    import torch
    from envidr_amd.encoding import get_encoder
    from tests.util import run_op
    enc, _ = get_encoder("hashgrid_diff", num_levels=4, log2_hashmap_size=12)
    enc = enc.cuda()
    enc.embeddings.data.uniform_(-1, 1)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.rand(200, 3, device="cuda", generator=g) * 1.8 - 0.9
    wgt = torch.rand(200, 8, device="cuda", generator=g)
    (enc(x) * wgt).sum().backward()
    got = enc.embeddings.grad
    assert got is not None and torch.isfinite(got).all() and got.abs().sum() > 0
    B, D, C, L = 200, 3, 2, 4
    S, Hres = float(np.log2(enc.per_level_scale)), int(enc.base_resolution)
    x01 = ((x + 1) / 2).cpu().numpy().astype(np.float32)
    grad = wgt.view(B, L, C).permute(1, 0, 2).contiguous().cpu().numpy()
    emb = enc.embeddings.detach().cpu().numpy()
    want = run_op("oracle", "hash_encode_backward", grad, x01, emb, enc.offsets.cpu().numpy(), np.zeros_like(emb), B, D, C, L, S, Hres, 0,
                  None, None)[4]
    assert rel_l2(got.cpu().numpy(), want) <= 2e-6
    # with the table frozen nothing is allocated or scattered (inference: normals only)
    enc.embeddings.requires_grad_(False)
    xi = x.clone().requires_grad_(True)
    enc(xi).sum().backward()
    assert xi.grad is not None and enc.embeddings.grad is got


@pytest.mark.parametrize("fused", [True, False])
def test_indirect_three_pass_matches_reference(fused):
    """BASELINE config #4 (use_renv + indir_ref): geometry pass -> reflected rays -> main pass with the
    renv branch (renderer.py:437-513 orchestration), against the reference's own three-pass render."""
    import torch
    model, opt = build_model(scenes.toaster_scene(shape=scenes.torus(), seed=3), indir_ref=True)
    g = np.load(GOLD / "frame_toaster_indir_40.npz")
    H, W = int(g["H"]), int(g["W"])
    ro, rd = scenes.camera_rays(H, W, theta=float(g["theta"]), phi=float(g["phi"]))
    res = model.render(torch.from_numpy(ro).cuda()[None], torch.from_numpy(rd).cuda()[None], staged=True, bg_color=1, perturb=False,
                       get_normal_image=True, env_rot_radian=None, fused=fused, max_steps=opt.max_steps, T_thresh=opt.T_thresh,
                       dt_gamma=opt.dt_gamma, early_stop_steps=-1)
    torch.cuda.synchronize()
    for key in KEYS:
        got = res[key].detach().cpu().numpy().reshape(H * W, -1)
        want = g[key].reshape(H * W, -1)
        err = rel_l2(got, want)
        assert err <= 1e-4, f"{key} (fused={fused}): rel-L2 {err:.3e}"


@pytest.mark.parametrize("fused", [True, False])
def test_chunked_indirect_render_matches_reference(fused):
    """`max_ray_batch_cuda` < N with indir_ref (reference renderer.py:428-436 chunks outside its three passes): the gather form of the three
    passes, every pass rendered in chunks of 500 of its rays and concatenated -- the fused geometry pass's chunk-local frame state must not
    reach the concatenation (round-4 advisor finding: torch.cat over the '_frame' dicts)"""
    import torch
    model, opt = build_model(scenes.toaster_scene(shape=scenes.torus(), seed=3), indir_ref=True)
    model.opt.max_ray_batch_cuda = 500
    g = np.load(GOLD / "frame_toaster_indir_40.npz")
    H, W = int(g["H"]), int(g["W"])
    ro, rd = scenes.camera_rays(H, W, theta=float(g["theta"]), phi=float(g["phi"]))
    res = model.render(torch.from_numpy(ro).cuda()[None], torch.from_numpy(rd).cuda()[None], staged=True, bg_color=1, perturb=False,
                       get_normal_image=True, env_rot_radian=None, fused=fused, max_steps=opt.max_steps, T_thresh=opt.T_thresh,
                       dt_gamma=opt.dt_gamma, early_stop_steps=-1)
    torch.cuda.synchronize()
    assert "_frame" not in res
    for key in KEYS:
        got = res[key].detach().cpu().numpy().reshape(H * W, -1)
        want = g[key].reshape(H * W, -1)
        err = rel_l2(got, want)
        assert err <= 1e-4, f"{key} (fused={fused}, chunked): rel-L2 {err:.3e}"


@pytest.mark.parametrize("fused", [True, False])
def test_indirect_with_an_object_box_matches_reference(fused):
    """`--obj_aabb` (reference renderer.py:91-97, 458-460): reflected rays are traced only from points inside the object's
    box.  The fixture's box cuts the torus -- 472 of the 1600 pixels differ from the frame without a box, by up to 0.03 -- and
    the frame must match what the reference rendered with the same option, in both forms of the three passes."""
    import torch
    g = np.load(GOLD / "frame_toaster_indir_aabb_40.npz")
    model, opt = build_model(scenes.toaster_scene(shape=scenes.torus(), seed=3), indir_ref=True, obj_aabb=[float(v) for v in g["obj_aabb"]])
    assert model.obj_aabb is not None and np.allclose(model.obj_aabb.cpu().numpy(), g["obj_aabb_scaled"])
    H, W = int(g["H"]), int(g["W"])
    ro, rd = scenes.camera_rays(H, W, theta=float(g["theta"]), phi=float(g["phi"]))
    res = model.render(torch.from_numpy(ro).cuda()[None], torch.from_numpy(rd).cuda()[None], staged=True, bg_color=1, perturb=False,
                       get_normal_image=True, env_rot_radian=None, fused=fused, max_steps=opt.max_steps, T_thresh=opt.T_thresh,
                       dt_gamma=opt.dt_gamma, early_stop_steps=-1)
    torch.cuda.synchronize()
    for key in KEYS:
        got = res[key].detach().cpu().numpy().reshape(H * W, -1)
        err = rel_l2(got, g[key].reshape(H * W, -1))
        assert err <= 1e-4, f"{key} (fused={fused}): rel-L2 {err:.3e}"
    # ... and it is the box that made the difference: without it this frame is the other fixture's
    plain = np.load(GOLD / "frame_toaster_indir_40.npz")
    assert rel_l2(res["image"].detach().cpu().numpy().reshape(H * W, -1), plain["image"].reshape(H * W, -1)) > 1e-3


@pytest.mark.parametrize("fused", [True, False])
def test_relight_with_shipped_checkpoints(fused):
    """README.md:136-146 relighting (`--sh_degree 4 --hidden_dim_env 160 --intensity_scale=0.8 --roughness_scale=0.8`):
    the scene is loaded through the checkpoint reader, the reference's shipped rendering MLPs and environment #3 are
    transplanted by `load_color_mlps` / `swap_env_path`, and the frame must match what the reference rendered with
    the same weights (fused kernel variant <4,5> and the operator loop)."""
    import torch
    from envidr_amd.nerf import checkpoint
    from tests import relight
    g = relight.fixture()
    scene = scenes.toaster_scene(hidden_env=160, ide_deg=4, seed=4)
    model, opt = build_model(scene, **relight.OVERRIDES)
    scene_ckpt = {"model": {k: v.clone() for k, v in model.state_dict().items()}}
    fresh, _ = build_model(scenes.toaster_scene(hidden_env=160, ide_deg=4, seed=9), **relight.OVERRIDES)
    info = checkpoint.load_checkpoint(fresh, scene_ckpt, swap_env_path=relight.shipped_state(g, "env"))
    assert not info["missing_keys"] and not info["unexpected_keys"]
    checkpoint.load_color_mlps(fresh, relight.shipped_state(g, "mlps"))
    assert fresh.supports_fused()
    H, W = int(g["H"]), int(g["W"])
    ro, rd = scenes.camera_rays(H, W, theta=float(g["theta"]), phi=float(g["phi"]))
    res = fresh.render(torch.from_numpy(ro).cuda()[None], torch.from_numpy(rd).cuda()[None], staged=True, bg_color=1, perturb=False,
                       get_normal_image=True, fused=fused, max_steps=opt.max_steps, T_thresh=opt.T_thresh, dt_gamma=opt.dt_gamma)
    torch.cuda.synchronize()
    for key in KEYS:
        got = res[key].detach().cpu().numpy().reshape(H * W, -1)
        err = rel_l2(got, g[key].reshape(H * W, -1))
        assert err <= 1e-4, f"{key} (fused={fused}): rel-L2 {err:.3e}"


LEGO = dict(scale=0.8, encoding_dir="sphere_harmonics", sh_degree=4, wo_viewdir=False, use_env_net=False, use_reflected_dir=False,
            diffuse_with_env=False, use_renv=False)            # tests/golden/lego_like.ini through the reference's parser


@pytest.mark.parametrize("fused", [True, False])
def test_config1_no_env_network(fused):
    """BASELINE configs[1]: hash-grid SDF + diffuse / specular MLPs with SH-encoded view direction and normal"""
    import torch
    model, opt = build_model(scenes.lego_scene(seed=8), **LEGO)
    assert model.color_net[0].weight.shape == (64, 45) and model.env_net is None
    g = np.load(GOLD / "frame_lego_48.npz")
    H, W = int(g["H"]), int(g["W"])
    ro, rd = scenes.camera_rays(H, W, theta=float(g["theta"]), phi=float(g["phi"]))
    res = model.render(torch.from_numpy(ro).cuda()[None], torch.from_numpy(rd).cuda()[None], staged=True, bg_color=1, perturb=False,
                       get_normal_image=True, fused=fused, max_steps=opt.max_steps, T_thresh=opt.T_thresh, dt_gamma=opt.dt_gamma)
    torch.cuda.synchronize()
    for key in KEYS:
        got = res[key].detach().cpu().numpy().reshape(H * W, -1)
        err = rel_l2(got, g[key].reshape(H * W, -1))
        assert err <= 1e-4, f"{key} (fused={fused}): rel-L2 {err:.3e}"


def test_fused_indirect_passes_equal_the_operator_loop():
    """each indirect pass on its own, fused kernel vs operator loop: geometry-only (normals composited, no shading) and
    the main pass with per-ray reflected radiance (renv MLP + second specular head + learnt blend), with visibilities on
    both sides of the 0.9 gate and roughness on both sides of indir_roughness_thresh"""
    import torch
    model, opt = build_model(scenes.toaster_scene(shape=scenes.torus(), seed=3), indir_ref=True, indir_roughness_thresh=0.06)
    ro, rd = (torch.from_numpy(a).cuda()[None] for a in scenes.camera_rays(56, 56, theta=40.0, phi=-50.0))
    kw = dict(bg_color=0, perturb=False, get_normal_image=True, max_steps=opt.max_steps, T_thresh=opt.T_thresh, dt_gamma=opt.dt_gamma)
    assert model.supports_fused(geometry_only=True)
    a = model._run(ro, rd, main_pass=False, geometry_only=True, fused=True, **kw)
    b = model._run(ro, rd, main_pass=False, geometry_only=True, fused=False, **kw)
    assert a["image"] is None and b["image"] is None
    for key in ("depth", "weights_sum", "normal_image"):
        assert rel_l2(a[key].cpu().numpy(), b[key].cpu().numpy()) <= 2e-5, key
    gen = torch.Generator().manual_seed(5)
    N = ro.shape[1]
    r_images = torch.cat([torch.rand(1, N, 3, generator=gen), (torch.rand(1, N, 1, generator=gen) > 0.4).float() * 0.97], -1).cuda()
    assert model.supports_fused(r_images=r_images)
    a = model._run(ro, rd, main_pass=True, r_images=r_images, fused=True, **kw)
    b = model._run(ro, rd, main_pass=True, r_images=r_images, fused=False, **kw)
    plain = model._run(ro, rd, main_pass=True, fused=True, **kw)
    for key in KEYS:
        x, y = a[key].cpu().numpy().reshape(N, -1), b[key].cpu().numpy().reshape(N, -1)
        assert rel_l2(x, y) <= 2e-5, f"{key}: {rel_l2(x, y):.3e}"
    # the branch did something: specular differs from the render without reflected radiance
    assert rel_l2(a["specular_image"].cpu().numpy(), plain["specular_image"].cpu().numpy()) > 1e-3


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fused_vs_operator_loop_on_random_scenes(seed):
    """randomised cross-check of the two render paths: random blobby occupancy, random (non-camera) rays including rays
    that start inside the box, random env rotation and render knobs; fused kernel vs the reference-shaped operator loop"""
    import torch
    rng = np.random.default_rng(100 + seed)
    centres = rng.uniform(-0.6, 0.6, size=(6, 3))
    radii = rng.uniform(0.12, 0.3, size=6)
    blobs = lambda p: (np.linalg.norm(p[:, None, :] - centres[None], axis=-1) < radii[None]).any(1)
    scene = scenes.toaster_scene(table_scale=float(rng.uniform(0.05, 0.3)), shape=blobs, sdf_bias=float(rng.uniform(-0.01, 0.02)),
                                 beta=float(rng.uniform(0.01, 0.05)), seed=50 + seed)
    knobs = dict(max_steps=int(rng.choice([256, 512, 1024])), T_thresh=float(rng.choice([1e-4, 1e-3])),
                 dt_gamma=float(rng.choice([0.0, 1 / 256])))
    model, opt = build_model(scene, **knobs, enabled_levels=int(rng.choice([-1, 10])), intensity_scale=float(rng.uniform(0.5, 1.0)))
    model.min_near = float(rng.choice([0.0, 0.2]))
    n = 3000 + 7 * seed
    o = rng.uniform(-1.6, 1.6, size=(n, 3))
    o[: n // 4] *= 0.4                                                       # some origins inside the scene box
    target = rng.uniform(-0.7, 0.7, size=(n, 3))
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ro, rd = torch.from_numpy(o.astype(np.float32)).cuda()[None], torch.from_numpy(d.astype(np.float32)).cuda()[None]
    kw = dict(staged=True, bg_color=float(rng.uniform(0, 1)), perturb=False, get_normal_image=True,
              env_rot_radian=float(rng.uniform(0, 6.28)), **knobs)
    assert model.supports_fused()
    a = model.render(ro, rd, fused=True, **kw)
    b = model.render(ro, rd, fused=False, **kw)
    torch.cuda.synchronize()
    assert float(b["weights_sum"].max()) > 0.5
    # The SDF network is piecewise linear: where a hidden pre-activation is within fp32 rounding of zero, two correct fp32
    # evaluations (different summation orders) take different sides of the ReLU kink and that SAMPLE's normal jumps.  Measured:
    # ~5e-6 of all samples.  So: all but a handful of rays agree to rounding, and the few others are bounded by one sample's weight.
    for key in KEYS:
        x, y = a[key].cpu().numpy().reshape(n, -1), b[key].cpu().numpy().reshape(n, -1)
        per_ray = np.abs(x - y).max(axis=1)
        flipped = per_ray > 1e-4
        assert flipped.sum() <= 3, f"seed {seed} {key}: {int(flipped.sum())} rays differ by more than 1e-4"
        assert rel_l2(x[~flipped], y[~flipped]) <= 3e-5, f"seed {seed} {key}: {rel_l2(x[~flipped], y[~flipped]):.3e}"
        assert rel_l2(x, y) <= 2e-3, f"seed {seed} {key}: {rel_l2(x, y):.3e}"


@pytest.mark.parametrize("ide_deg,hidden", [(4, 160), (5, 128), (4, 128)])
def test_indirect_frames_of_the_other_built_shapes_equal_the_operator_loop(ide_deg, hidden):
    """the reflected-radiance instantiations of the record-shading kernel for the (IDE degree, hidden width) shapes the reference
    fixtures do not cover (tools/kernel_coverage.sh listed them as never launched): three-pass frame, fused against the operator
    loop (torch MLPs, the reference's statements)"""
    import torch
    model, opt = build_model(scenes.toaster_scene(shape=scenes.torus(), seed=3, hidden_env=hidden, ide_deg=ide_deg), indir_ref=True,
                             sh_degree=ide_deg, hidden_dim_env=hidden)
    assert model.supports_fused()
    ro, rd = scenes.camera_rays(36, 36, theta=50.0, phi=-30.0)
    frames = {}
    for fused in (True, False):
        res = model.render(torch.from_numpy(ro).cuda()[None], torch.from_numpy(rd).cuda()[None], staged=True, bg_color=1, perturb=False,
                           get_normal_image=True, env_rot_radian=None, fused=fused, max_steps=opt.max_steps, T_thresh=opt.T_thresh,
                           dt_gamma=opt.dt_gamma, early_stop_steps=-1)
        torch.cuda.synchronize()
        frames[fused] = {k: res[k].detach().cpu().numpy().reshape(36 * 36, -1) for k in KEYS}
    assert frames[True]["weights_sum"].max() > 0.9          # the frame sees the object (and its reflected rays)
    for key in KEYS:
        err = rel_l2(frames[True][key], frames[False][key])
        assert err <= 1e-4, f"{key}: rel-L2 {err:.3e}"
