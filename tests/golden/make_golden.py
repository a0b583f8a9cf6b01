"""Generate the committed golden vectors from the REFERENCE ITSELF, run on the CPU in the build
container (the only place /root/reference exists).  Never runs on the GPU box.

    python tests/golden/make_golden.py

What runs (SURVEY.md 8c levels 1 and 2):
  * the reference's own Python -- `nerf.options.config_parser()` on configs/scenes/toaster.ini,
    `NeRFNetwork(...)`, `forward_sigma` / `get_color_mlp_extra_params` / `forward_color`,
    `IntegratedDirEncoder`, and `model.render(...)` -> `run_cuda` inference loop -- imported
    unmodified from /root/reference;
  * with its extension modules (`raymarching._ext._raymarching`, `hashencoder._ext._hashencoder`,
    ...) played by oracle/_ref: the reference's kernel bodies executed on the CPU.

Harness-side adaptations only (no reference file is touched): third-party packages that are not
installed and not on the path (trimesh, cv2, ...) are stubbed in sys.modules, `np.math` is restored
(numpy 2 dropped it), `Tensor.cuda()` is the identity, and a small argparse-based stand-in plays
`configargparse`.

Model parameters are NOT stored: both this script and the tests regenerate them from
`envidr_amd.scenes.toaster_scene(seed)` (numpy PCG64, platform independent) and this script
injects them into the reference model.  The fixtures hold inputs + the reference's outputs.
"""
from __future__ import annotations

import argparse
import math
import sys
import types
from pathlib import Path
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
REFERENCE = Path("/root/reference")
OUT = Path(__file__).resolve().parent

from envidr_amd import scenes  # noqa: E402
from oracle import clib  # noqa: E402

F = np.float32


# ------------------------------------------------------------------------------------------------
# harness: make the reference importable and runnable on the CPU
# ------------------------------------------------------------------------------------------------
class _IniArgumentParser(argparse.ArgumentParser):
    """argparse + `--config file.ini` (key = value | [a, b] | True) expanded into argv."""

    def add_argument(self, *a, **kw):
        kw.pop("is_config_file", None)
        return super().add_argument(*a, **kw)

    def parse_args(self, args=None, namespace=None):
        argv = list(sys.argv[1:] if args is None else args)
        out = []
        i = 0
        while i < len(argv):
            if argv[i] == "--config":
                out += self._expand(argv[i + 1])
                i += 2
            else:
                out.append(argv[i])
                i += 1
        # command line wins over the file: file args first
        cfg = [a for a in out if isinstance(a, tuple)]
        cli = [a for a in out if not isinstance(a, tuple)]
        flat = [x for t in cfg for x in t] + cli
        return super().parse_args(flat, namespace)

    @staticmethod
    def _expand(path):
        res = []
        for raw in Path(path).read_text().splitlines():
            line = raw.split("#")[0].split(";")[0].strip()
            if not line or "=" not in line:
                continue
            key, val = (s.strip() for s in line.split("=", 1))
            if val == "True":
                res.append((f"--{key}",))
            elif val in ("False", ""):
                continue
            elif val.startswith("["):
                items = [v.strip() for v in val.strip("[]").split(",") if v.strip()]
                res.append((f"--{key}", *items))
            else:
                res.append((f"--{key}", val))
        return res


class _RefBackend:
    """Plays a pybind extension module: same function names / argument orders as the reference's
    bindings, taking CPU torch tensors, forwarding to oracle/_ref through the shared C signature."""

    def __init__(self, names):
        self._ref = clib.ref()
        for n in names:
            setattr(self, n, self._make(n))

    def _make(self, name):
        from envidr_amd._lib import SIGNATURES
        sig = SIGNATURES[name]

        def fn(*args):
            conv = []
            for kind, a in zip(sig, args):
                if kind == "p":
                    if a is None:
                        conv.append(None)
                    else:
                        assert a.is_contiguous(), name
                        conv.append(a.detach().numpy())
                else:
                    conv.append(a)
            self._ref.call(name, *conv)
        return fn


def install_reference():
    np.math = math
    torch.Tensor.cuda = lambda self, *a, **k: self
    for mod in ["trimesh", "cv2", "tensorboardX", "mcubes", "torch_ema", "lpips", "imageio", "cachetools", "open3d",
                "open3d.visualization", "open3d.visualization.rendering", "torchvision", "torchvision.transforms",
                "dearpygui", "dearpygui.dearpygui", "IPython", "matplotlib", "matplotlib.pyplot", "packaging"]:
        if mod not in sys.modules:
            try:
                __import__(mod)
            except Exception:
                sys.modules[mod] = MagicMock()
    cap = types.ModuleType("configargparse")
    cap.ArgumentParser = _IniArgumentParser
    sys.modules["configargparse"] = cap

    rm = _RefBackend(["near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "get_scatter_idx",
                      "march_rays_train", "composite_rays_train_forward", "composite_rays_train_backward", "march_rays",
                      "composite_rays"])
    he = _RefBackend(["hash_encode_forward", "hash_encode_backward", "hash_encode_second_backward"])
    sh = _RefBackend(["sh_encode_forward", "sh_encode_backward"])
    ge = _RefBackend(["grid_encode_forward", "grid_encode_backward"])
    for pkg, name, backend in [("raymarching", "_raymarching", rm), ("hashencoder", "_hashencoder", he), ("shencoder", "_shencoder", sh),
                               ("gridencoder", "_gridencoder", ge)]:
        ext = types.ModuleType(f"{pkg}._ext")
        setattr(ext, name, backend)
        sys.modules[f"{pkg}._ext"] = ext
        sys.modules[f"{pkg}._ext.{name}"] = backend
    sys.path.insert(0, str(REFERENCE))


def build_reference_model(scene: scenes.SceneParams, extra_argv=(), config=None):
    """the reference's own option parser + constructor (main_nerf.py:16,59-78), then our seeded weights."""
    from nerf.options import config_parser
    from nerf.network import NeRFNetwork
    old = sys.argv
    sys.argv = ["main_nerf.py", "--config", str(config or REFERENCE / "configs/scenes/toaster.ini"), "--test", *extra_argv]
    try:
        opt = config_parser()
    finally:
        sys.argv = old
    model = NeRFNetwork(encoding="hashgrid", encoding_dir=opt.encoding_dir, bound=opt.bound, cuda_ray=opt.cuda_ray,
                        density_scale=1, min_near=opt.min_near, density_thresh=opt.density_thresh, bg_radius=opt.bg_radius,
                        use_sdf=opt.use_sdf, hidden_dim=opt.hidden_dim, num_layers=opt.num_layers,
                        num_layers_color=opt.num_layers_color, hidden_dim_color=opt.hidden_dim_color,
                        num_layers_bg=opt.num_layers_bg, num_levels=opt.num_levels, geo_feat_dim=opt.geo_feat_dim,
                        opt=opt, env_opt=None)
    model.eval()
    with torch.no_grad():
        assert tuple(model.encoder.embeddings.shape) == scene.table.shape
        assert np.array_equal(model.encoder.offsets.numpy(), scene.offsets)
        model.encoder.embeddings.data = torch.from_numpy(scene.table.copy())
        for name, attr in [("sdf", "sdf_net"), ("env", "env_net"), ("diffuse", "diffuse_net"), ("specular", "color_net"),
                           ("renv", "renv_net")]:
            net = getattr(model, attr, None)
            if net is None:
                assert name not in scene.mlps, name
                continue
            assert len(net) == len(scene.mlps[name]), name
            for lin, (W, b) in zip(net, scene.mlps[name]):
                assert tuple(lin.weight.shape) == W.shape, (name, lin.weight.shape, W.shape)
                lin.weight.data = torch.from_numpy(W.copy())
                lin.bias.data = torch.from_numpy(b.copy())
        model.sdf_density.beta.data = torch.tensor(scene.beta)
        if model.cuda_ray:                        # (no occupancy grid without cuda_ray)
            model.density_bitfield.data = torch.from_numpy(scene.bitfield.copy())
    return model, opt


# ------------------------------------------------------------------------------------------------
# fixtures
# ------------------------------------------------------------------------------------------------
def sample_points(rng, n):
    """points around the r = 0.5 shell with unit view directions, plus the degenerate ones"""
    p = rng.normal(size=(n, 3))
    p = p / np.linalg.norm(p, axis=1, keepdims=True) * rng.uniform(0.44, 0.56, size=(n, 1))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    p[0] = 0.0          # the padded-sample position the reference also shades
    p[1] = (1.0, -1.0, 1.0)
    return p.astype(F), d.astype(F)


def golden_shading(model, opt, tag, env_rot=None, n=1536, seed=5):
    rng = np.random.default_rng(seed)
    xyz, dirs = sample_points(rng, n)
    x = torch.from_numpy(xyz).requires_grad_(True)
    d = torch.from_numpy(dirs)
    sdfs, sigmas, geo, normals, _ = model.forward_sigma(x, use_sdf_sigma_grad=True, dirs=d, dists=None)
    rough = model.roughness
    blend = model.blend_weight
    n_enc, w_r_enc, n_dot, n_env_enc = model.get_color_mlp_extra_params(normals, d, rough, env_rot)
    rgb = model.forward_color(geo, d, n_enc, w_r_enc, n_dot, True, n_env_enc=n_env_enc, r_images=None, roughness=rough)
    g = lambda t: np.zeros(0, F) if t is None else t.detach().numpy().astype(F)
    np.savez_compressed(OUT / f"shading_{tag}.npz", xyz=xyz, dirs=dirs, env_rot=np.array(np.nan if env_rot is None else env_rot),
                        sdf=g(sdfs), sigma=g(sigmas), geo_feat=g(geo), normal=g(normals), roughness=g(rough), blend=g(blend),
                        w_r_enc=g(w_r_enc), n_env_enc=g(n_env_enc), n_dot=g(n_dot), n_enc=g(n_enc), c_diffuse=g(model.c_diffuse),
                        c_specular=g(model.c_specular), rgb=g(rgb))
    print(f"[golden] shading_{tag}: {n} samples, mean sigma {float(sigmas.mean()):.2f}, mean roughness {float(rough.mean()):.4f}")


def golden_frame(model, opt, tag, H, W, env_rot=None, theta=30.0, phi=-20.0, extra=None):
    import nerf.render_func.cuda_ray as cuda_ray
    ro, rd = scenes.camera_rays(H, W, theta=theta, phi=phi)
    trace = []
    orig = cuda_ray.raymarching.march_rays

    def traced(n_alive, n_step, *a, **k):   # record the (n_alive, n_step) schedule the reference ran
        out = orig(n_alive, n_step, *a, **k)
        trace.append((int(n_alive), int(n_step), int(out[0].shape[0]), int((out[2][:, 0] > 0).sum())))
        return out

    cuda_ray.raymarching.march_rays = traced
    try:
        res = model.render(torch.from_numpy(ro)[None], torch.from_numpy(rd)[None], staged=True, bg_color=1, perturb=False,
                           get_normal_image=True, env_rot_radian=env_rot, **vars(opt))
    finally:
        cuda_ray.raymarching.march_rays = orig
    g = lambda t: t.detach().numpy().astype(F).reshape(H * W, -1).squeeze(-1) if t.numel() == H * W else t.detach().numpy().astype(F).reshape(H * W, -1)
    np.savez_compressed(OUT / f"frame_{tag}.npz", H=H, W=W, theta=theta, phi=phi,
                        env_rot=np.array(np.nan if env_rot is None else env_rot),
                        image=g(res["image"]), depth=g(res["depth"]), weights_sum=g(res["weights_sum"]),
                        normal_image=g(res["normal_image"]), diffuse_image=g(res["diffuse_image"]),
                        specular_image=g(res["specular_image"]), roughness_image=g(res["roughness_image"]),
                        trace=np.array(trace, np.int32), **(extra or {}))
    print(f"[golden] frame_{tag}: {H}x{W}, {len(trace)} loop iterations, {sum(t[3] for t in trace)} samples, "
          f"hit fraction {float((res['weights_sum'] > 0).float().mean()):.3f}, mean rgb {res['image'].mean(dim=(0, 1)).tolist()}")


def golden_instance_norm():
    """geo_feat_act = env_feat_act = instanceNorm (reference network.py:436-440, 542-546, 601-605): the per-sample chain and a frame"""
    model, opt = build_reference_model(scenes.toaster_scene(seed=2), extra_argv=["--geo_feat_act", "instanceNorm", "--env_feat_act", "instanceNorm"])
    assert opt.geo_feat_act == "instanceNorm" and opt.env_feat_act == "instanceNorm"
    golden_shading(model, opt, "toaster_inorm", n=1024)
    golden_frame(model, opt, "toaster_inorm_32", 32, 32, theta=75.0, phi=-30.0)


OBJ_AABB = [-1.5, -1.5, -0.12, 0.35, 1.5, 1.5]      # in dataset units (x scale 0.65): cuts through the torus


def golden_indirect_aabb():
    """config #4 with `--obj_aabb`: reflected rays are only traced from points inside the object's box (reference
    renderer.py:91-97, 458-460); the box cuts the torus so that both sides of the mask occur among the hit pixels"""
    scene4 = scenes.toaster_scene(shape=scenes.torus(), seed=3)
    model, opt = build_reference_model(scene4, extra_argv=["--indir_ref", "--obj_aabb", *map(str, OBJ_AABB)])
    assert opt.indir_ref and model.obj_aabb is not None
    golden_frame(model, opt, "toaster_indir_aabb_40", 40, 40, theta=40.0, phi=-50.0,
                 extra={"obj_aabb": np.array(OBJ_AABB, F), "obj_aabb_scaled": model.obj_aabb.numpy().astype(F)})


def golden_background():
    """run_cuda's background-sphere branch (cuda_ray.py:56-62, network.py:343-367,727-742): toaster.ini + `--bg_radius 3`.
    The background model's parameters (2-D hash grid table, bias-free MLP) are the reference's own seeded initialisation
    scaled up so that the background is not flat; they are stored in the fixture as the test's input data."""
    scene = scenes.toaster_scene(seed=6)
    model, opt = build_reference_model(scene, extra_argv=["--bg_radius", "3"])
    assert model.bg_radius == 3 and model.bg_net is not None
    g = torch.Generator().manual_seed(17)
    with torch.no_grad():
        model.encoder_bg.embeddings.data = (torch.rand(model.encoder_bg.embeddings.shape, generator=g) * 2 - 1) * 0.5
        for lin in model.bg_net:
            lin.weight.data = (torch.rand(lin.weight.shape, generator=g) * 2 - 1) * (6.0 / lin.weight.shape[1]) ** 0.5
    extra = {"bg_table": model.encoder_bg.embeddings.detach().numpy().astype(F), "bg_offsets": model.encoder_bg.offsets.numpy(),
             "bg_radius": np.float32(3.0)}
    for i, lin in enumerate(model.bg_net):
        extra[f"bg_w{i}"] = lin.weight.detach().numpy().astype(F)
    H = W = 40
    ro, rd = scenes.camera_rays(H, W, theta=300.0, phi=-15.0)
    res = model.render(torch.from_numpy(ro)[None], torch.from_numpy(rd)[None], staged=True, bg_color=None, perturb=False,
                       get_normal_image=True, env_rot_radian=None, **vars(opt))
    g2 = lambda t: t.detach().numpy().astype(F).reshape(H * W, -1).squeeze(-1) if t.numel() == H * W else t.detach().numpy().astype(F).reshape(H * W, -1)
    np.savez_compressed(OUT / "frame_toaster_bg_40.npz", H=H, W=W, theta=300.0, phi=-15.0, image=g2(res["image"]), depth=g2(res["depth"]),
                        weights_sum=g2(res["weights_sum"]), sphere_bg=g2(res["sphere_bg"]), **extra)
    print(f"[golden] frame_toaster_bg_40: sphere_bg mean {res['sphere_bg'].mean(0).tolist()}, std {float(res['sphere_bg'].std()):.3f}, "
          f"hit fraction {float((res['weights_sum'] > 0).float().mean()):.3f}")


def golden_relight():
    """README.md:136-146 relighting: toaster.ini + `--sh_degree 4 --hidden_dim_env 160 --intensity_scale=0.8
    --roughness_scale=0.8`, with the weights the reference SHIPS: ckpts/rendering_mlps.pth (diffuse / specular /
    renv MLPs, loaded the way utils.py:509-530 does) and ckpts/env_ckpts/env_net_3.pth as the swapped-in
    environment.  Geometry (table, sdf_net, bitfield) is the seeded synthetic scene.  The shipped tensors are
    stored in the fixture under their original key names: they are the test's input data."""
    scene = scenes.toaster_scene(hidden_env=160, ide_deg=4, seed=4)
    model, opt = build_reference_model(scene, extra_argv=["--sh_degree", "4", "--hidden_dim_env", "160", "--intensity_scale=0.8",
                                                          "--roughness_scale=0.8"])
    mlps = torch.load(REFERENCE / "ckpts/rendering_mlps.pth", map_location="cpu")["model"]
    env = torch.load(REFERENCE / "ckpts/env_ckpts/env_net_3.pth", map_location="cpu")["model"]
    sub = lambda prefix: {k[len(prefix):]: v for k, v in mlps.items() if k.startswith(prefix)}
    model.color_net.load_state_dict(sub("color_net."))
    model.diffuse_net.load_state_dict(sub("diffuse_net."))
    model.renv_net.load_state_dict(sub("renv_net."))
    # extract_env_ckpt (sph_loader.py:372) writes 'env_net' + '0.weight': the ModuleList index follows directly
    model.env_net.load_state_dict({k[len("env_net"):]: v for k, v in env.items()})
    extra = {f"mlps/{k}": v.numpy() for k, v in mlps.items()}
    extra.update({f"env/{k}": v.numpy() for k, v in env.items()})
    golden_frame(model, opt, "relight_40", 40, 40, theta=75.0, phi=-25.0, extra=extra)


def golden_demo(res=200):
    """BASELINE configs[0]: the reference's own CPU-runnable case, demo.ipynb cell 17 (one surface sample per ray
    on the unit sphere, constant hash feature, IDE degree 4, env MLP 38-160-160-160-12 twice, diffuse / specular
    heads) executed from the notebook's code cells as they are, with the shipped demo/ weights.  Only the image
    size is changed (800 -> `res`, intrinsics recomputed by the notebook's own formula).  The weights are stored
    in the fixture as input data; the 400x400 mean-RGB anchors of SURVEY.md section 6 are recomputed and stored too."""
    import json
    import os
    nb = json.load(open(REFERENCE / "demo.ipynb"))
    code = {i: "".join(c["source"]) for i, c in enumerate(nb["cells"]) if c["cell_type"] == "code"}
    plt = MagicMock()
    np.math = math
    for mod in ("matplotlib", "matplotlib.pyplot"):
        if mod not in sys.modules:
            try:
                __import__(mod)
            except Exception:
                sys.modules[mod] = MagicMock()
    cwd = os.getcwd()
    os.chdir(REFERENCE)
    try:
        def run(size):
            env = {"__name__": "demo"}
            for i in (1, 3, 5):
                exec(code[i], env)
            env["plt"] = plt
            exec(f"W, H = {size}, {size}\nfocal = W / (2 * np.tan(camera_angle_x / 2))\nintrinsics = np.array([focal, focal, W/2, H/2])", env)
            for i in (7, 9, 10, 13, 14):
                exec(code[i], env)
            exec(code[16], env)                               # the rendering parameters (theta, phi, material, env)
            exec(code[17].split("out_img = {}")[0], env)      # rendering steps, without the plotting tail
            return env
        env = run(400)
        mean400 = env["image"].mean(0).numpy()
        print("[golden] demo 400x400 mean rgb", mean400.tolist(), "(SURVEY.md section 6 anchor: 0.62849, 0.70200, 0.82242)")
        env = run(res)
    finally:
        os.chdir(cwd)
    out = {"res": res, "theta": env["theta"], "phi": env["phi"], "radius": env["radius"], "roughness": env["roughness"],
           "metallic": env["metallic"], "base_color": np.array(env["base_color"], F), "mean_rgb_400": mean400.astype(F),
           "xyz_encoding": env["xyz_encoding"].numpy(), "image": env["image"].numpy(), "diffuse": env["diffuse"].numpy(),
           "specular": env["specular"].numpy(), "mask": env["mask"].numpy(), "kappa_inv": env["kappa_inv"].numpy()}
    for name in ("sdf_net", "env_net", "diffuse_net", "specular_net"):
        for k, v in env[name].state_dict().items():
            out[f"{name}/{k}"] = v.numpy()
    np.savez_compressed(OUT / "demo_sphere.npz", **out)
    print(f"[golden] demo_sphere: {res}x{res}, {int(env['mask'].sum())} hit rays, mean rgb {env['image'].mean(0).tolist()}")


GRID_SCENE = dict(table_scale=0.3, sdf_bias=0.0, beta=0.05, seed=6)
GRID_POSES = [(20.0, -30.0), (140.0, -10.0), (260.0, -60.0)]


def grid_poses():
    return np.stack([scenes.nerf_matrix_to_ngp(scenes.pose_spherical(th, ph, 4.0), scale=0.65) for th, ph in GRID_POSES])


def golden_grid():
    """occupancy-grid maintenance run by the reference (renderer.py:200-359): mark_untrained_grid on three
    cameras, two full updates (the second one exercises the decayed running max), one partial update.
    Random jitter comes from torch's CPU generator seeded with 21; the GPU test replays that stream."""
    scene = scenes.toaster_scene(**GRID_SCENE)
    model, opt = build_reference_model(scene)
    model.density_grid.zero_()
    model.density_bitfield.zero_()
    out = {}
    stride = 61

    def snap(tag):
        grid = model.density_grid.numpy()
        out[f"{tag}/grid_sample"] = grid.reshape(-1)[::stride].astype(F).copy()
        out[f"{tag}/bitfield"] = model.density_bitfield.numpy().copy()
        out[f"{tag}/mean_density"] = np.float64(model.mean_density)
        out[f"{tag}/n_negative"] = np.int64((grid < 0).sum())
        out[f"{tag}/n_above_10"] = np.int64((grid > 10).sum())
        print(f"[golden] grid {tag}: mean density {model.mean_density:.4f}, untrained {int((grid < 0).sum())}, "
              f"occupied bits {int(np.unpackbits(model.density_bitfield.numpy()).sum())}")

    torch.manual_seed(21)
    model.mark_untrained_grid(grid_poses(), scenes.intrinsics_for(800, 800))
    snap("marked")
    model.update_extra_state()
    snap("full1")
    model.update_extra_state()
    snap("full2")
    model.iter_density = 16
    model.update_extra_state()
    snap("partial")
    np.savez_compressed(OUT / "grid_update.npz", stride=stride, **out)


def golden_rays():
    """the reference's get_rays (nerf/utils.py:109-209, full-image branch) on a non-square image and two poses"""
    from nerf.utils import get_rays
    poses = np.stack([scenes.nerf_matrix_to_ngp(scenes.pose_spherical(33.0, -27.0, 4.0), scale=0.65),
                      scenes.nerf_matrix_to_ngp(scenes.pose_spherical(250.0, -70.0, 3.0), scale=0.8)]).astype(F)
    H, W = 12, 20
    intr = np.array([31.5, 29.0, 10.25, 5.75])
    r = get_rays(torch.from_numpy(poses), intr, H, W, -1)
    np.savez_compressed(OUT / "get_rays.npz", poses=poses, intrinsics=intr, H=H, W=W, rays_o=r["rays_o"].numpy(), rays_d=r["rays_d"].numpy())
    print("[golden] get_rays:", tuple(r["rays_d"].shape))


def golden_ide():
    from ide_encoder.ide_encoder import IntegratedDirEncoder
    rng = np.random.default_rng(7)
    out = {}
    for deg in (4, 5):
        enc = IntegratedDirEncoder(3, deg)
        d = rng.normal(size=(2000, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        d = d.astype(F)
        d[0] = (0, 0, 1)
        d[1] = (0, 0, -1)
        rough = rng.uniform(0.0, 0.2, size=(2000, 1)).astype(F)
        rough[:50] = 0
        out[f"dirs{deg}"] = d
        out[f"rough{deg}"] = rough
        out[f"ide{deg}_rough"] = enc(torch.from_numpy(d), torch.from_numpy(rough)).numpy()
        out[f"ide{deg}_k064"] = enc(torch.from_numpy(d), 0.64).numpy()
        out[f"mat{deg}"] = enc.mat.numpy()
    np.savez_compressed(OUT / "ide.npz", **out)
    print("[golden] ide: deg 4 and 5, per-sample and scalar roughness")
    # the gradient torch autograd takes through the reference's forward (what its training branch back-propagates): upstream
    # gradient = a fixed random matrix; roughness in the range the roughness head produces (the l = 16 terms of the reference's
    # fp32 polynomial are pure rounding noise near |z| = 1: with these roughness values they are attenuated by > e^-4)
    out = {}
    for deg in (4, 5):
        enc = IntegratedDirEncoder(3, deg)
        d = rng.normal(size=(1500, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        d = d.astype(F)
        d[0] = (0.6, 0.8, 0.0)
        rough = rng.uniform(0.03, 0.3, size=(1500, 1)).astype(F)
        gout = rng.normal(size=(1500, enc.output_dim)).astype(F)
        dt, rt = torch.from_numpy(d).requires_grad_(True), torch.from_numpy(rough).requires_grad_(True)
        (enc(dt, rt) * torch.from_numpy(gout)).sum().backward()
        out.update({f"dirs{deg}": d, f"rough{deg}": rough, f"gout{deg}": gout, f"gdirs{deg}": dt.grad.numpy(), f"grough{deg}": rt.grad.numpy()})
        dt2 = torch.from_numpy(d).requires_grad_(True)
        (enc(dt2, 0.64) * torch.from_numpy(gout)).sum().backward()
        out[f"gdirs{deg}_k064"] = dt2.grad.numpy()
    np.savez_compressed(OUT / "ide_grad.npz", **out)
    print("[golden] ide_grad: deg 4 and 5, autograd of the reference's forward")


def golden_ops():
    """outputs of the reference kernel bodies on a few operator cases, so the GPU box (which has
    neither /root/reference nor necessarily oracle/_ref) can still check against reference data."""
    from tests import cases
    from tests.util import bits_equal, run_op
    from envidr_amd._lib import SIGNATURES
    keep = {"march/c1_step8", "march/c2_bound2", "march/c1_cone_noise", "composite/rgb", "hash/D3C2L16_fwd_grad",
            "hash/D2C1L5_fwd_grad", "grid/D3C2L16g0a0_fwd_grad", "near_far/special", "misc/packbits", "misc/morton",
            "freq_sh/sh_deg4_grad", "freq_sh/freq_D3deg4", "train/composite_train_bwd", "hash_bwd/D3C2L8_bwd2"}
    out = {}
    for cid, op, args, tol in cases.all_cases():
        if cid not in keep:
            continue
        res = run_op("ref", op, *args)
        ptr_args = [a for kind, a in zip(SIGNATURES[op], args) if kind == "p"]
        for k, (r, a) in enumerate(zip(res, ptr_args)):
            # keep only what the call produced (arrays it changed), not the inputs it was given
            if r is not None and r.size and not bits_equal(r, np.ascontiguousarray(a)):
                out[f"{cid.replace('/', '.')}|{k}"] = r
    np.savez_compressed(OUT / "ops_ref.npz", **out)
    print(f"[golden] ops_ref: {len(out)} arrays from {len(keep)} cases")


def golden_torch_only():
    """Fixtures from the reference's TORCH-ONLY code: no kernel body, no keyword header, no oracle/_ref anywhere in the arithmetic.

    * `encoding.FreqEncoder` (encoding.py:6-44, the `frequency_torch` branch of get_encoder: max_freq_log2 = multires - 1,
      N_freqs = multires) -- forward and, through torch autograd, the input gradient: pins freq_encode_forward / _backward.
    * the volume-rendering arithmetic of `nerf/render_func/non_cuda_ray.run` (non_cuda_ray.py:108-156: deltas, alphas, the cumprod
      transmittance, weights, weights_sum, normal image, depth, image + background), executed by calling the reference's own `run`
      on a model object of ours whose `density` / `color` return prescribed tensors.  The one extension call in that function,
      `raymarching.near_far_from_aabb`, is answered by the harness with prescribed (near, far): they are inputs of the fixture.
      Pins the compositing recurrence of composite_rays / composite_rays_train_forward.
    """
    import encoding as ref_encoding
    rng = np.random.default_rng(41)
    out = {}
    for D, deg in ((3, 4), (3, 10), (2, 6), (1, 1)):
        enc = ref_encoding.FreqEncoder(input_dim=D, max_freq_log2=deg - 1, N_freqs=deg, log_sampling=True)
        x = rng.uniform(-1.0, 1.0, size=(700, D)).astype(F)
        x[0] = 0
        x[1] = 1
        x[2] = -1
        xt = torch.from_numpy(x).requires_grad_(True)
        y = enc(xt)
        g = rng.normal(size=tuple(y.shape)).astype(F)
        (y * torch.from_numpy(g)).sum().backward()
        tag = f"freq_D{D}deg{deg}"
        out.update({f"{tag}|x": x, f"{tag}|y": y.detach().numpy(), f"{tag}|g": g, f"{tag}|gx": xt.grad.numpy()})
        assert y.shape[1] == D + 2 * D * deg
    print("[golden] torch_only: FreqEncoder", [k for k in out if k.endswith("|y")])

    from nerf.render_func import non_cuda_ray
    N, T = 300, 48
    nears = rng.uniform(0.3, 2.0, size=N).astype(F)
    fars = (nears + rng.uniform(0.5, 2.5, size=N)).astype(F)
    sigma = rng.uniform(0, 1, size=(N, T)).astype(F) ** 4 * 300            # mostly thin, some opaque
    sigma[:20] = 0                                                          # rays through empty space
    sigma[20:40, 10:] = 1e4                                                 # rays that saturate (T underflows to the 1e-15 floor)
    rgb = rng.uniform(0, 1, size=(N, T, 3)).astype(F)
    normal = rng.normal(size=(N, T, 3)).astype(F)
    bg = np.array([0.25, 0.5, 0.75], F)
    rays_o = rng.normal(size=(N, 3)).astype(F)
    rays_d = rng.normal(size=(N, 3)).astype(F)
    rays_d /= np.linalg.norm(rays_d, axis=1, keepdims=True)

    class _Opt:
        debug = False
        backsdf_loss = False
        eikonal_loss = False

    class _Model:
        opt = _Opt()
        use_normal_with_mlp = True
        use_n_dot_viewdir = True
        use_reflected_dir = True
        training = False
        aabb_train = aabb_infer = torch.tensor([-8.0, -8, -8, 8, 8, 8])
        min_near = 0.2
        density_scale = 1
        bg_radius = -1

        def density(self, xyzs, **kw):
            return {"sigma": torch.from_numpy(sigma).reshape(-1, 1), "normal": torch.from_numpy(normal).reshape(-1, 3)}

        def color(self, xyzs, dirs, mask=None, **kw):
            return torch.from_numpy(rgb).reshape(-1, 3)

    class _NearFar:                                     # stands in for the `raymarching` module inside non_cuda_ray only
        @staticmethod
        def near_far_from_aabb(rays_o, rays_d, aabb, min_near):
            return torch.from_numpy(nears.copy()), torch.from_numpy(fars.copy())

    saved = non_cuda_ray.raymarching
    non_cuda_ray.raymarching = _NearFar
    try:
        res = non_cuda_ray.run(_Model(), torch.from_numpy(rays_o), torch.from_numpy(rays_d), num_steps=T, upsample_steps=0,
                               bg_color=torch.from_numpy(bg), perturb=False, get_normal_image=True)
    finally:
        non_cuda_ray.raymarching = saved
    out.update({"vr|nears": nears, "vr|fars": fars, "vr|sigma": sigma, "vr|rgb": rgb, "vr|normal": normal, "vr|bg": bg,
                "vr|image": res["image"].detach().numpy(), "vr|depth": res["depth"].detach().numpy(),
                "vr|weights_sum": res["weights_sum"].detach().numpy(), "vr|normal_image": res["normal_image"].detach().numpy()})
    np.savez_compressed(OUT / "torch_only.npz", **out)
    print("[golden] torch_only: volume rendering", tuple(res["image"].shape), "weights_sum",
          float(res["weights_sum"].min()), "...", float(res["weights_sum"].max()))


def resample_stub(torch_mod):
    """the analytic stand-in model of golden_torch_only_resample / tests/test_plain_cpu.py: density and colour are functions of the sample
    POSITION (a soft shell of radius 1.2), so that the importance re-sampling and the merge of the two sample sets matter"""
    torch = torch_mod

    class _Opt:
        debug = False
        backsdf_loss = False
        eikonal_loss = False

    class _Model:
        opt = _Opt()
        use_normal_with_mlp = True
        use_n_dot_viewdir = True
        use_reflected_dir = True
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

    return _Model()


def golden_torch_only_resample():
    """The reference's `non_cuda_ray.run` WITH importance re-sampling (utils.py sample_pdf, the sort / gather merge of non_cuda_ray.py:71-106)
    on the analytic stand-in model above -- torch on the CPU on both sides, no kernel anywhere: pins envidr_amd/nerf/render_func/
    non_cuda_ray.py (inverse_cdf_samples, the merge, the compositing) in tests/test_plain_cpu.py."""
    from nerf.render_func import non_cuda_ray
    rng = np.random.default_rng(43)
    N = 200
    rays_o = (rng.normal(size=(N, 3)) * 0.2 + np.array([0.0, 0.0, -3.0])).astype(F)
    rays_d = (rng.normal(size=(N, 3)) * 0.25 + np.array([0.0, 0.0, 1.0])).astype(F)
    rays_d /= np.linalg.norm(rays_d, axis=1, keepdims=True)
    nears = rng.uniform(0.8, 1.4, size=N).astype(F)
    fars = (nears + rng.uniform(2.5, 3.5, size=N)).astype(F)
    bg = np.array([0.1, 0.2, 0.3], F)

    class _NearFar:
        @staticmethod
        def near_far_from_aabb(rays_o, rays_d, aabb, min_near):
            return torch.from_numpy(nears.copy()), torch.from_numpy(fars.copy())

    saved = non_cuda_ray.raymarching
    non_cuda_ray.raymarching = _NearFar
    out = {"rays_o": rays_o, "rays_d": rays_d, "nears": nears, "fars": fars, "bg": bg}
    try:
        for tag, steps, up in (("a", 40, 24), ("b", 24, 40), ("c", 64, 0)):
            res = non_cuda_ray.run(resample_stub(torch), torch.from_numpy(rays_o), torch.from_numpy(rays_d), num_steps=steps, upsample_steps=up,
                                   bg_color=torch.from_numpy(bg), perturb=False, get_normal_image=True)
            out.update({f"{tag}|steps": np.array([steps, up], np.int32), f"{tag}|image": res["image"].detach().numpy(),
                        f"{tag}|depth": res["depth"].detach().numpy(), f"{tag}|weights_sum": res["weights_sum"].detach().numpy(),
                        f"{tag}|normal_image": res["normal_image"].detach().numpy()})
            print(f"[golden] torch_only_resample {tag}: {steps}+{up}, weights_sum {float(res['weights_sum'].min()):.3f} ... {float(res['weights_sum'].max()):.3f}")
    finally:
        non_cuda_ray.raymarching = saved
    np.savez_compressed(OUT / "torch_only_resample.npz", **out)


SPH_BETA = 0.005
SPH_MATERIAL = {"roughness": 0.3, "metallic": 0.2, "color": [20 / 255, 70 / 255, 160 / 255, 1.0]}
SPH_ENV_INDEX = 3


def build_reference_sph_model():
    """the reference's NeRFNetwork as main_nerf.py:47-78 builds it for configs/neural_renderer.ini (env_sph_mode: run_sph, the SDF
    network with the material parameters concatenated, one environment MLP per environment), with the weights the reference SHIPS --
    demo/sdf_net.pth (37-64-64-14), ckpts/rendering_mlps.pth (diffuse / specular heads), ckpts/env_ckpts/env_net_3.pth -- and the
    hash table of envidr_amd.scenes.sphere_table (demo/xyz_encoding.txt + seeded noise).  `env_opt` is what nerf/sph_loader.py's
    config_parser returns, minus the dataset paths (sph_loader imports Open3D, which is not installed): the three vary_* flags of
    configs/env_dataset_config.ini and the names of configs/ktx_images_list.txt."""
    from nerf.options import config_parser
    from nerf.network import NeRFNetwork
    old = sys.argv
    sys.argv = ["main_nerf.py", "--config", str(REFERENCE / "configs/neural_renderer.ini"), "--test"]
    try:
        opt = config_parser()
    finally:
        sys.argv = old
    assert opt.env_sph_mode and not opt.cuda_ray
    names = (REFERENCE / "configs/ktx_images_list.txt").read_text().splitlines()
    env_opt = types.SimpleNamespace(vary_roughness=True, vary_metallic=True, vary_base_color=True, env_images_names=names)
    model = NeRFNetwork(encoding="hashgrid", encoding_dir=opt.encoding_dir, bound=opt.bound, cuda_ray=opt.cuda_ray, density_scale=1,
                        min_near=opt.min_near, density_thresh=opt.density_thresh, bg_radius=opt.bg_radius, use_sdf=opt.use_sdf,
                        hidden_dim=opt.hidden_dim, num_layers=opt.num_layers, num_layers_color=opt.num_layers_color,
                        hidden_dim_color=opt.hidden_dim_color, num_layers_bg=opt.num_layers_bg, num_levels=opt.num_levels,
                        geo_feat_dim=opt.geo_feat_dim, opt=opt, env_opt=env_opt)
    model.eval()
    sdf = torch.load(REFERENCE / "demo/sdf_net.pth", map_location="cpu")
    mlps = torch.load(REFERENCE / "ckpts/rendering_mlps.pth", map_location="cpu")["model"]
    env = torch.load(REFERENCE / "ckpts/env_ckpts/env_net_3.pth", map_location="cpu")["model"]
    xyz_encoding = np.loadtxt(REFERENCE / "demo/xyz_encoding.txt").astype(F)
    sub = lambda prefix: {k[len(prefix):]: v for k, v in mlps.items() if k.startswith(prefix)}
    with torch.no_grad():
        for i, j in enumerate((0, 2, 4)):          # demo/sdf_net.pth is an nn.Sequential with ReLUs between the Linears
            model.sdf_net[i].weight.data, model.sdf_net[i].bias.data = sdf[f"{j}.weight"], sdf[f"{j}.bias"]
        model.color_net.load_state_dict(sub("color_net."))
        model.diffuse_net.load_state_dict(sub("diffuse_net."))
        model.env_nets[SPH_ENV_INDEX].load_state_dict({k[len("env_net"):]: v for k, v in env.items()})
        model.encoder.embeddings.data = torch.from_numpy(scenes.sphere_table(xyz_encoding))
        model.sdf_density.beta.data = torch.tensor(SPH_BETA)
    opt.env_sph_radius = 0.95 * opt.scale         # main_nerf.py:100,132: the dataset's sphere radius, in the scaled scene
    shipped = {f"sdf/{k}": v.numpy() for k, v in sdf.items()}
    shipped.update({f"mlps/{k}": v.numpy() for k, v in mlps.items() if k.startswith(("color_net.", "diffuse_net."))})
    shipped.update({f"env/{k}": v.numpy() for k, v in env.items()})
    shipped["xyz_encoding"] = xyz_encoding
    return model, opt, shipped


def golden_sph():
    """BASELINE configs[0]'s lineage as the reference itself renders it: `model.render()` -> run_sph (12 samples around the analytic
    hit, material-conditioned SDF network, torch compositing) at 200 x 200 without and at 40 x 40 with the normal image, plus the
    `staged` form (chunks of 4096 rays, each normalising its depth by its own largest far).  The [N,N,3] normal image the reference
    returns here (renderer.py:539-540 broadcasts weights_sum [N,1] against [1,N,3]) is stored as its diagonal: the per-ray blend."""
    model, opt, shipped = build_reference_sph_model()
    kw = {k: v for k, v in vars(opt).items() if k not in ("bg_color", "perturb", "material", "env_net_index")}
    out = dict(shipped, beta=SPH_BETA, env_net_index=SPH_ENV_INDEX, radius=opt.env_sph_radius, scale=opt.scale,
               material=np.array([SPH_MATERIAL["roughness"], SPH_MATERIAL["metallic"], *SPH_MATERIAL["color"][:3]], F))
    for tag, res, normal, staged in (("200", 200, False, False), ("40", 40, True, False), ("64s", 64, False, True)):
        ro, rd = scenes.camera_rays(res, res, theta=123.0, phi=10.0, radius=4.0, scale=opt.scale)
        r = model.render(torch.from_numpy(ro)[None], torch.from_numpy(rd)[None], staged=staged, bg_color=1, perturb=False,
                         get_normal_image=normal, env_net_index=SPH_ENV_INDEX, material=dict(SPH_MATERIAL), **kw)
        N = res * res
        out.update({f"{tag}|res": res, f"{tag}|theta": 123.0, f"{tag}|phi": 10.0,
                    f"{tag}|image": r["image"].detach().numpy().reshape(N, 3), f"{tag}|depth": r["depth"].detach().numpy().reshape(N),
                    f"{tag}|diffuse_image": r["diffuse_image"].detach().numpy().reshape(N, 3),
                    f"{tag}|specular_image": r["specular_image"].detach().numpy().reshape(N, 3)})
        if not staged:
            out[f"{tag}|weights_sum"] = r["weights_sum"].detach().numpy().reshape(N)
            out[f"{tag}|sigmas"] = r["sigmas"].detach().numpy()
            out[f"{tag}|sdfs"] = r["sdfs"].detach().numpy()
        if normal:
            ni = r["normal_image"].detach()
            assert tuple(ni.shape) == (N, N, 3)
            out[f"{tag}|normal_image"] = ni[torch.arange(N), torch.arange(N)].numpy()
        ws = r["weights_sum"].reshape(-1) if "weights_sum" in r else None
        print(f"[golden] sph {tag}: {res}x{res}" + ("" if ws is None else f", {int((ws > 0).sum())} hit rays, weights_sum {float(ws[ws > 0].min()):.3f}"
              f" ... {float(ws.max()):.3f}") + f", mean rgb {r['image'].reshape(-1, 3).mean(0).tolist()}")
    np.savez_compressed(OUT / "sph_render.npz", **out)


def golden_variants():
    """forward_geometry's other forms, which no shipped config selects (reference network.py:46-102, 153-222, 417): `--use_neus_sdf`
    (NeuS section alphas instead of the Laplace density; the compositors then take input_alpha), `--geometric_init` (weight-normalised
    layers, Softplus(beta = 100)) and `--skip_layers 1` (layer 1 takes cat([h, x]) / sqrt 2) -- all three on, toaster.ini otherwise:
    a shading chain and a 32 x 32 frame through run_cuda.  The SDF network's weights are seeded here (shapes 32-32, 64-64, 64-15) and
    installed as weight_v with weight_g = the row norms, i.e. as the effective weights."""
    from nerf.options import config_parser
    from nerf.network import NeRFNetwork
    scene = scenes.toaster_scene(seed=9)
    rng = np.random.default_rng(77)
    scene.mlps["sdf"] = [scenes.xavier_linear(rng, 32, 32), scenes.xavier_linear(rng, 64, 64), scenes.xavier_linear(rng, 64, 15)]
    scene.mlps["sdf"][-1][1][0] = 0.005
    for _, b in scene.mlps["sdf"][:2]:
        b += rng.uniform(-0.05, 0.05, size=b.shape).astype(F)
    old = sys.argv
    sys.argv = ["main_nerf.py", "--config", str(REFERENCE / "configs/scenes/toaster.ini"), "--test", "--use_neus_sdf", "--geometric_init",
                "--skip_layers", "1", "--init_variance", "0.45"]
    try:
        opt = config_parser()
    finally:
        sys.argv = old
    model = NeRFNetwork(encoding="hashgrid", encoding_dir=opt.encoding_dir, bound=opt.bound, cuda_ray=opt.cuda_ray, density_scale=1,
                        min_near=opt.min_near, density_thresh=opt.density_thresh, bg_radius=opt.bg_radius, use_sdf=opt.use_sdf,
                        hidden_dim=opt.hidden_dim, num_layers=opt.num_layers, num_layers_color=opt.num_layers_color,
                        hidden_dim_color=opt.hidden_dim_color, num_layers_bg=opt.num_layers_bg, num_levels=opt.num_levels,
                        geo_feat_dim=opt.geo_feat_dim, opt=opt, env_opt=None)
    model.eval()
    with torch.no_grad():
        model.encoder.embeddings.data = torch.from_numpy(scene.table.copy())
        for lin, (W, b) in zip(model.sdf_net, scene.mlps["sdf"]):
            assert tuple(lin.weight_v.shape) == W.shape, (lin.weight_v.shape, W.shape)
            lin.weight_v.data = torch.from_numpy(W.copy())
            lin.weight_g.data = torch.from_numpy(np.linalg.norm(W, axis=1, keepdims=True).astype(F))
            lin.bias.data = torch.from_numpy(b.copy())
        for name, attr in [("env", "env_net"), ("diffuse", "diffuse_net"), ("specular", "color_net"), ("renv", "renv_net")]:
            for lin, (W, b) in zip(getattr(model, attr), scene.mlps[name]):
                lin.weight.data, lin.bias.data = torch.from_numpy(W.copy()), torch.from_numpy(b.copy())
        model.density_bitfield.data = torch.from_numpy(scene.bitfield.copy())
    # the shading chain: forward_sigma needs the step sizes for the NeuS alphas
    rng = np.random.default_rng(5)
    xyz, dirs = sample_points(rng, 1024)
    dists = np.full(1024, 2 * math.sqrt(3) / 1024, F)
    x = torch.from_numpy(xyz).requires_grad_(True)
    d = torch.from_numpy(dirs)
    sdfs, alphas, geo, normals, _ = model.forward_sigma(x, use_sdf_sigma_grad=True, dirs=d, dists=torch.from_numpy(dists))
    n_enc, w_r_enc, n_dot, n_env_enc = model.get_color_mlp_extra_params(normals, d, model.roughness, None)
    rgb = model.forward_color(geo, d, n_enc, w_r_enc, n_dot, True, n_env_enc=n_env_enc, r_images=None, roughness=model.roughness)
    g = lambda t: t.detach().numpy().astype(F)
    out = {"xyz": xyz, "dirs": dirs, "dists": dists, "sdf": g(sdfs), "alpha": g(alphas), "geo_feat": g(geo), "normal": g(normals),
           "roughness": g(model.roughness), "rgb": g(rgb), "init_variance": np.float32(0.45)}
    for i, (W, b) in enumerate(scene.mlps["sdf"]):
        out[f"sdf/{i}.weight"], out[f"sdf/{i}.bias"] = W, b
    np.savez_compressed(OUT / "shading_variants.npz", **out)
    print(f"[golden] shading_variants: mean alpha {float(alphas.mean()):.4f}, sdf {float(sdfs.min()):.3f} ... {float(sdfs.max()):.3f}")
    golden_frame(model, opt, "variants_32", 32, 32, theta=60.0, phi=-25.0)


def train_targets(n):
    """deterministic stand-in for ground-truth pixels of a training batch"""
    i = np.arange(n, dtype=np.float64)
    return np.stack([0.5 + 0.3 * np.sin(0.37 * i), 0.5 + 0.3 * np.cos(0.11 * i + 1.0), 0.4 + 0.2 * np.sin(0.05 * i + 2.0)], -1).astype(F)


def train_loss(res, target):
    """a scalar that reads every differentiable output of run_cuda's training branch the reference's Trainer reads
    (nerf/utils.py:700-800: colour, eikonal, back-sdf style terms, mask), with fixed weights; the GPU test uses this very function"""
    loss = ((res["image"].reshape(-1, 3) - target) ** 2).mean()
    loss = loss + 0.05 * ((res["sdf_gradients"].norm(p=2, dim=-1) - 1) ** 2).mean()
    loss = loss + 0.02 * ((res["relsdf"] - res["est_relsdf"]).abs() * res["sdf_weights"]).sum() / max(res["relsdf"].shape[0], 1)
    loss = loss + 0.1 * (res["weights_sum"].reshape(-1) - 0.7).abs().mean() + 0.03 * res["depth"].reshape(-1).mean()
    return loss


def golden_train(tag, scene, config=None, H=32, W=32, theta=55.0, phi=-30.0):
    """One training-mode forward + backward of the REFERENCE: model.train(), render() -> run_cuda's training branch
    (cuda_ray.py:64-237: march_rays_train -> forward_sigma with autograd normals -> forward_color -> composite_rays_train),
    eikonal + back-sdf switches on, loss = train_loss, .backward() through the reference's own autograd Functions
    (_hash_encode second-order backward, _composite_rays_train backward) on the reference kernel bodies.  Stored: the
    forward outputs, the per-ray sample counts, and the gradients of every MLP parameter, beta and the hash table
    (table: per-level norms, touched-row counts and a sample of rows)."""
    model, opt = build_reference_model(scene, config=config)
    model.train()
    opt.eikonal_loss, opt.backsdf_loss = True, True
    ro, rd = scenes.camera_rays(H, W, theta=theta, phi=phi)
    N = H * W
    target = torch.from_numpy(train_targets(N))
    kw = dict(vars(opt))
    res = model.render(torch.from_numpy(ro)[None], torch.from_numpy(rd)[None], staged=False, bg_color=1, perturb=False, force_all_rays=False, **kw)
    loss = train_loss(res, target)
    model.zero_grad()
    loss.backward()
    out = {"H": H, "W": W, "theta": theta, "phi": phi, "loss": np.float64(loss.item()), "image": res["image"].detach().numpy().reshape(N, 3),
           "depth": res["depth"].detach().numpy().reshape(N), "weights_sum": res["weights_sum"].detach().numpy().reshape(N),
           "n_samples": np.int64(res["sigmas"].shape[0]), "n_relsdf": np.int64(res["relsdf"].shape[0]),
           "counter": model.step_counter[0].numpy().copy()}
    nets = ["sdf_net", "diffuse_net", "color_net"] + (["env_net"] if getattr(model, "env_net", None) is not None else [])
    for name in nets:
        for k, prm in getattr(model, name).named_parameters():
            g = prm.grad.numpy().astype(F)
            out[f"norm/{name}.{k}"] = np.float64(np.linalg.norm(g.astype(np.float64)))
            out[f"grad/{name}.{k}"] = g if (name != "env_net" or g.ndim == 1) else g.reshape(-1)[::7].copy()
    out["grad/beta"] = model.sdf_density.beta.grad.numpy().astype(F)
    ge = model.encoder.embeddings.grad.numpy()
    offs = model.encoder.offsets.numpy()
    out["emb/level_norm"] = np.array([np.linalg.norm(ge[offs[l]:offs[l + 1]].astype(np.float64)) for l in range(16)])
    touched = np.nonzero(np.any(ge != 0, axis=1))[0]
    out["emb/level_touched"] = np.array([int(((touched >= offs[l]) & (touched < offs[l + 1])).sum()) for l in range(16)], np.int64)
    pick = touched[:: max(1, touched.size // 4096)][:4096]
    out["emb/rows"], out["emb/values"] = pick.astype(np.int64), ge[pick].astype(F)
    np.savez_compressed(OUT / f"train_{tag}.npz", **out)
    print(f"[golden] train_{tag}: {H}x{W}, {int(out['n_samples'])} samples, loss {loss.item():.6f}, |grad sdf_net.0.weight| "
          f"{out['norm/sdf_net.0.weight']:.4e}, table rows touched {touched.size}, beta grad {float(out['grad/beta']):.4e}")


def golden_non_cuda_ray():
    """The reference's torch-only render function (nerf/render_func/non_cuda_ray.py `run`, cuda_ray = False) on the plainest SDF configuration
    (tests/golden/plain_like.ini: the function hands the colour network neither a reflected direction nor n.v nor an encoded normal): a
    16x16 view with uniform samples only, and one with importance re-sampling."""
    scene = scenes.plain_scene()
    model, opt = build_reference_model(scene, config=OUT / "plain_like.ini")
    assert not opt.cuda_ray and not model.cuda_ray
    H = W = 16
    ro, rd = scenes.camera_rays(H, W, theta=60.0, phi=-25.0, scale=0.8)
    kw = {k: v for k, v in vars(opt).items() if k not in ("num_steps", "upsample_steps", "max_ray_batch")}
    out = {}
    for tag, steps, up in (("uniform", 96, 0), ("resampled", 64, 32)):
        res = model.render(torch.from_numpy(ro)[None], torch.from_numpy(rd)[None], staged=False, bg_color=1, perturb=False,
                           get_normal_image=True, num_steps=steps, upsample_steps=up, **kw)
        for k in ("image", "depth", "weights_sum", "normal_image"):
            out[f"{tag}|{k}"] = res[k].detach().numpy().astype(F).reshape(H * W, -1)
        out[f"{tag}|steps"] = np.array([steps, up], np.int32)
        print(f"[golden] non_cuda_ray {tag}: {steps}+{up} samples per ray, hit fraction {float((res['weights_sum'] > 0.5).float().mean()):.3f}, "
              f"mean rgb {res['image'].mean(dim=(0, 1)).tolist()}")
    # (staged=True cannot be taken with this function: render() then reads results_['roughness_image'] & co for every name in visual_items,
    #  which `run` never returns, and the option parser does not accept an empty visual_items -- renderer.py:407-412)
    np.savez_compressed(OUT / "frame_plain_nocuda_16.npz", H=H, W=W, theta=60.0, phi=-25.0, scale=0.8, **out)


# tag -> (lines removed from torch_like.ini, lines added): the variants of tests/test_network_cpu.py (its NETWORK_VARIANTS holds the same
# settings as RenderOptions overrides)
NETWORK_CPU_VARIANTS = {
    "sdf": ([], []),
    "density": (["use_sdf = True"], []),
    "neus": ([], ["use_neus_sdf = True"]),
    "skip": (["num_layers = 3"], ["num_layers = 4", "skip_layers = [2]"]),
    "geoinit": ([], ["geometric_init = True"]),
    "tanh_separate_roughness": (["geo_feat_act = unitNorm", "ensemble_mlp = True", "learn_indir_blend = True"], ["geo_feat_act = tanh"]),
    "instance_norm_detached_annealed": (["geo_feat_act = unitNorm"], ["geo_feat_act = instanceNorm", "detach_normal = True", "normal_anneal_ratio = 0.5"]),
    "diffuse_only": ([], ["diffuse_only = True"]),
    # the toaster.ini structure with the identity in place of the integrated-direction encoding (encoding_ref = frequency, zero frequencies):
    # reflected direction -> environment network -> colour network, diffuse side through the same network; env rotation; intensity scales
    "env": ([], ["use_reflected_dir = True", "use_env_net = True", "diffuse_with_env = True", "wo_viewdir = True", "hidden_dim_env = 24",
                 "light_intensity_scale = 1.3", "intensity_scale = 0.9"]),
    "env_add": ([], ["use_reflected_dir = True", "use_env_net = True", "diffuse_with_env = True", "diffuse_env_fusion = add", "env_feat_dim = 12",
                     "hidden_dim_env = 24", "env_feat_act = tanh"]),
    "env_mul_split": ([], ["use_reflected_dir = True", "use_env_net = True", "diffuse_with_env = True", "diffuse_env_fusion = mul", "env_feat_dim = 12",
                           "split_diffuse_env = True", "hidden_dim_env = 24", "hidden_dim_env_diffuse = 20", "env_wo_bias = True"]),
    "env_no_diffuse_env": ([], ["use_reflected_dir = True", "use_env_net = True", "hidden_dim_env = 24", "env_feat_act = instanceNorm"]),
    "renv": ([], ["use_reflected_dir = True", "use_env_net = True", "diffuse_with_env = True", "wo_viewdir = True", "hidden_dim_env = 24", "use_renv = True",
                  "indir_roughness_thresh = 0.12"]),
    "renv_fixed_blend": (["learn_indir_blend = True"], ["use_reflected_dir = True", "use_env_net = True", "diffuse_with_env = True", "hidden_dim_env = 24",
                                                        "use_renv = True", "indir_roughness_thresh = 0.12"]),
}
# how forward_color is driven per variant (tests/test_network_cpu.py NETWORK_CALLS holds the same): env rotation, reflected radiance
NETWORK_CPU_CALLS = {"env": {"env_rot": 0.7}, "env_mul_split": {"env_rot": -1.9}, "renv": {"r_images": 4, "env_rot": 0.4}, "renv_fixed_blend": {"r_images": 3}}


def golden_network_cpu():
    """The reference's NeRFNetwork in configurations that run in plain torch on the CPU (tests/golden/torch_like.ini: identity position and
    direction encoders, no integrated-direction encoding), with its OWN initial weights: the SDF family, the plain-density branch
    (`use_sdf` off: trunc_exp density, normals = the negated density gradient, network.py:424-429,519), the NeuS section alpha, the geometric
    initialisation with a skip layer, the other feature activations, a separate roughness layer, detached / annealed normals, diffuse only.
    Per sample: density / sdf, geometry feature, normal, roughness, colours; and the gradients of a scalar of them w.r.t. every parameter and
    the positions (through the normals: a double backward).  Pins envidr_amd/nerf/network.py without any kernel: tests/test_network_cpu.py."""
    import tempfile
    from nerf.options import config_parser
    from nerf.network import NeRFNetwork
    rng = np.random.default_rng(47)
    x = rng.uniform(-0.8, 0.8, size=(200, 3)).astype(F)
    d = rng.normal(size=(200, 3)).astype(F)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dists = rng.uniform(0.005, 0.02, size=200).astype(F)
    w_rgb, w_sigma = rng.normal(size=(200, 3)).astype(F), rng.normal(size=200).astype(F)
    r_images = rng.uniform(0, 1, size=(200, 4)).astype(F)
    r_images[:, 3] = (rng.uniform(size=200) < 0.7).astype(F)                   # visibility: 0 or 1
    out = {"x": x, "d": d, "dists": dists, "w_rgb": w_rgb, "w_sigma": w_sigma, "r_images": r_images}
    base = (OUT / "torch_like.ini").read_text()
    for tag, (drop, add) in NETWORK_CPU_VARIANTS.items():
        text = base
        for line in drop:
            assert line + "\n" in text, line
            text = text.replace(line + "\n", "")
        text += "".join(line + "\n" for line in add)
        with tempfile.NamedTemporaryFile("w", suffix=".ini", delete=False) as f:
            f.write(text)
        old = sys.argv
        sys.argv = ["main_nerf.py", "--config", f.name, "--test"]
        try:
            opt = config_parser()
        finally:
            sys.argv = old
        assert not opt.cuda_ray
        torch.manual_seed(3)
        model = NeRFNetwork(encoding="hashgrid", encoding_dir=opt.encoding_dir, bound=opt.bound, cuda_ray=opt.cuda_ray, density_scale=1,
                            min_near=opt.min_near, density_thresh=opt.density_thresh, bg_radius=opt.bg_radius, use_sdf=opt.use_sdf,
                            hidden_dim=opt.hidden_dim, num_layers=opt.num_layers, num_layers_color=opt.num_layers_color,
                            hidden_dim_color=opt.hidden_dim_color, num_layers_bg=opt.num_layers_bg, num_levels=opt.num_levels,
                            geo_feat_dim=opt.geo_feat_dim, opt=opt, env_opt=None)
        model.train()
        if not opt.geometric_init:
            with torch.no_grad():                   # (xavier weights of a 3-wide first layer give a nearly flat field: scale the geometry up)
                for lin in model.sdf_net:
                    lin.weight.mul_(2.5)
                    lin.bias.add_(0.1 * torch.randn_like(lin.bias))
        xt = torch.from_numpy(x).requires_grad_(True)
        dt = torch.from_numpy(d)
        sdfs, sigmas, geo, normals, _ = model.forward_sigma(xt, use_sdf_sigma_grad=True, dirs=dt, dists=torch.from_numpy(dists))
        rough = model.roughness
        call = NETWORK_CPU_CALLS.get(tag, {})
        n_enc, w_r, n_dot, n_env = model.get_color_mlp_extra_params(normals, dt, rough, call.get("env_rot"))
        ri = torch.from_numpy(r_images[:, :call["r_images"]].copy()) if "r_images" in call else None
        rgb = model.forward_color(geo, dt, n_enc, w_r, n_dot, True, n_env_enc=n_env, r_images=ri, roughness=rough)
        loss = (rgb * torch.from_numpy(w_rgb)).sum() + (sigmas.reshape(-1) * torch.from_numpy(w_sigma)).sum()
        params = dict(model.named_parameters())
        grads = torch.autograd.grad(loss, [xt, *params.values()], allow_unused=True)
        g = lambda t: np.zeros(0, F) if (t is None or not torch.is_tensor(t)) else t.detach().numpy().astype(F)
        out.update({f"{tag}|sdf": g(sdfs), f"{tag}|sigma": g(sigmas), f"{tag}|geo_feat": g(geo), f"{tag}|normal": g(normals), f"{tag}|roughness": g(rough),
                    f"{tag}|rgb": g(rgb), f"{tag}|c_diffuse": g(model.c_diffuse), f"{tag}|c_specular": g(model.c_specular), f"{tag}|grad|x": g(grads[0])})
        for (name, p), gr in zip(params.items(), grads[1:]):
            out[f"{tag}|param|{name}"] = g(p)
            out[f"{tag}|grad|{name}"] = g(gr)
        print(f"[golden] network_cpu {tag}: {len(params)} parameters, sigma {float(sigmas.min()):.3g} ... {float(sigmas.max()):.3g}, "
              f"|normal| {float(normals.norm(dim=-1).mean()):.3f}, mean rgb {[round(v, 4) for v in rgb.mean(0).tolist()]}")
    np.savez_compressed(OUT / "network_cpu.npz", **out)


SPH_CPU_LINES = ["env_sph_mode = True", "use_reflected_dir = True", "use_env_net = True", "diffuse_with_env = True", "wo_viewdir = True", "hidden_dim_env = 24",
                 "roughness_act_scale = 1.0"]
SPH_CPU_MATERIAL = {"roughness": 0.35, "metallic": 0.6, "color": [0.8, 0.5, 0.3]}
SPH_CPU_RADIUS = 0.76


def sph_cpu_weights(model):
    """weights for the env-sphere pin: the SDF network answers |x| - radius up to a small learnt wobble (layer 1 = +-x, +-y, +-z scaled, the
    rest seeded noise), so that the shell samples straddle a real surface; everything else stays the reference's own initialisation"""
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for lin in model.sdf_net:
            lin.weight.mul_(2.0)
            lin.bias.add_(0.05 * torch.randn(lin.bias.shape, generator=g))
        model.sdf_net[-1].bias[0] = -0.2
        model.sdf_density.beta.fill_(0.02)


def golden_sph_cpu():
    """The reference's `run_sph` (nerf/render_func/sph_ray.py:34-221) in plain torch on the CPU: tests/golden/torch_like.ini + env_sph_mode (the
    material parameters concatenated to the SDF input, one environment MLP per environment, identity encoders), 90 rays of which a third
    miss the sphere, 12 shell samples per hit.  Evaluation mode with the normal image; training mode with the perturbation off and every
    training extra on (backsdf_loss, eikonal_loss, sdf_loss_weight: relsdf / sdf_weights / sdf_dist / sdf_gradients / surf_sdfs), plus the
    gradients of a scalar of image, depth and the extras w.r.t. every parameter; and a call in which no ray hits.  Pins
    envidr_amd/nerf/render_func/sph_ray.py's operator form without any kernel: tests/test_sph_cpu.py."""
    import tempfile
    from nerf.options import config_parser
    from nerf.network import NeRFNetwork
    from nerf import render_func
    text = (OUT / "torch_like.ini").read_text().replace("visual_items = [roughness]", "visual_items = [roughness, diffuse, specular]")
    assert "diffuse, specular" in text
    text += "".join(l + "\n" for l in SPH_CPU_LINES)
    with tempfile.NamedTemporaryFile("w", suffix=".ini", delete=False) as f:
        f.write(text)
    old = sys.argv
    sys.argv = ["main_nerf.py", "--config", f.name, "--test"]
    try:
        opt = config_parser()
    finally:
        sys.argv = old
    assert opt.env_sph_mode and not opt.cuda_ray
    opt.env_sph_radius = SPH_CPU_RADIUS
    env_opt = types.SimpleNamespace(vary_roughness=True, vary_metallic=True, vary_base_color=True, env_images_names=["a", "b", "c"])
    torch.manual_seed(5)
    model = NeRFNetwork(encoding="hashgrid", encoding_dir=opt.encoding_dir, bound=opt.bound, cuda_ray=opt.cuda_ray, density_scale=1,
                        min_near=opt.min_near, density_thresh=opt.density_thresh, bg_radius=opt.bg_radius, use_sdf=opt.use_sdf,
                        hidden_dim=opt.hidden_dim, num_layers=opt.num_layers, num_layers_color=opt.num_layers_color,
                        hidden_dim_color=opt.hidden_dim_color, num_layers_bg=opt.num_layers_bg, num_levels=opt.num_levels,
                        geo_feat_dim=opt.geo_feat_dim, opt=opt, env_opt=env_opt)
    sph_cpu_weights(model)
    rng = np.random.default_rng(61)
    n = 90
    ro = rng.normal(size=(n, 3)).astype(F)
    ro *= (rng.uniform(1.8, 3.5, size=(n, 1)) / np.linalg.norm(ro, axis=1, keepdims=True)).astype(F)         # (origins at different distances: chunks then differ in their largest far)
    aim = rng.normal(size=(n, 3)).astype(F) * 0.35                      # two thirds aim into the sphere ...
    aim[::3] = aim[::3] / np.linalg.norm(aim[::3], axis=1, keepdims=True) * 1.6          # ... a third past it
    rd = aim - ro
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    w_img, w_depth = rng.normal(size=(n, 3)).astype(F), rng.normal(size=n).astype(F)
    out = {"rays_o": ro, "rays_d": rd, "w_img": w_img, "w_depth": w_depth, "radius": np.float32(SPH_CPU_RADIUS),
           "material": np.array([SPH_CPU_MATERIAL["roughness"], SPH_CPU_MATERIAL["metallic"], *SPH_CPU_MATERIAL["color"]], F)}
    for name, prm in model.named_parameters():
        out[f"param|{name}"] = prm.detach().numpy().copy()
    g = lambda v: np.zeros(0, F) if (v is None or not torch.is_tensor(v)) else v.detach().numpy().astype(F)
    o, d = torch.from_numpy(ro)[None], torch.from_numpy(rd)[None]

    # evaluation mode, normal image, second environment
    model.eval()
    opt.backsdf_loss = opt.eikonal_loss = False
    r = render_func.run_sph(model, o, d, bg_color=1, perturb=False, get_normal_image=True, env_net_index=1, material=dict(SPH_CPU_MATERIAL))
    for k, v in r.items():
        out[f"eval|{k}"] = g(v)
    hit = int((r["weights_sum"].reshape(-1) > 0).sum())
    print(f"[golden] sph_cpu eval: {hit} of {n} rays composite something, weights_sum up to {float(r['weights_sum'].max()):.3f}, keys {sorted(r)}")

    # the same through render(): one call with the normal image blended by weights_sum (renderer.py:539-540 broadcasts weights_sum [N,1]
    # against [1,N,3]: its diagonal is the per-ray blend), and staged in chunks of 32 rays (each normalises its depth by its own largest far)
    kw = {k: v for k, v in vars(opt).items() if k not in ("bg_color", "perturb", "material", "env_net_index", "max_ray_batch")}
    r = model.render(o, d, staged=False, bg_color=1, perturb=False, get_normal_image=True, env_net_index=1, material=dict(SPH_CPU_MATERIAL), **kw)
    ni = r["normal_image"].detach()
    assert tuple(ni.shape) == (n, n, 3)
    out["render|normal_image"] = ni[torch.arange(n), torch.arange(n)].numpy()
    out["render|image"], out["render|depth"] = g(r["image"]), g(r["depth"])
    r = model.render(o, d, staged=True, max_ray_batch=32, bg_color=1, perturb=False, get_normal_image=False, env_net_index=1,
                     material=dict(SPH_CPU_MATERIAL), **kw)
    for k, v in r.items():
        out[f"staged|{k}"] = g(v)
    print(f"[golden] sph_cpu render(): staged keys {sorted(r)}, depth differs from the one-call depth by up to "
          f"{float((r['depth'].reshape(-1) - torch.from_numpy(out['render|depth']).reshape(-1)).abs().max()):.3g}")

    # training mode with every extra
    model.train()
    opt.backsdf_loss = opt.eikonal_loss = True
    opt.sdf_loss_weight = 0.1
    r = render_func.run_sph(model, o, d, bg_color=1, perturb=False, env_net_index=2, material=dict(SPH_CPU_MATERIAL))
    for k, v in r.items():
        out[f"train|{k}"] = g(v)
    loss = ((r["image"][0] * torch.from_numpy(w_img)).sum() + (r["depth"][0] * torch.from_numpy(w_depth)).sum() + 0.3 * r["surf_sdfs"].abs().mean()
            + 0.2 * (r["relsdf"] * r["sdf_weights"] * r["sdf_dist"]).sum() + 0.1 * ((r["sdf_gradients"].norm(dim=-1) - 1) ** 2).mean())
    params = dict(model.named_parameters())
    grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    for (name, _), gr in zip(params.items(), grads):
        out[f"train|grad|{name}"] = g(gr)
    out["train|loss"] = np.float32(loss.item())
    print(f"[golden] sph_cpu train: loss {loss.item():.5f}, keys {sorted(r)}, parameters without gradient: {[k for (k, _), gr in zip(params.items(), grads) if gr is None]}")

    # no ray hits
    model.eval()
    opt.backsdf_loss = opt.eikonal_loss = False
    far_o = torch.from_numpy(ro[:5] * 4)[None]
    side = np.cross(ro[:5], np.array([0.3, -0.2, 0.9], F)).astype(F)
    side /= np.linalg.norm(side, axis=1, keepdims=True)
    r = render_func.run_sph(model, far_o, torch.from_numpy(side)[None], bg_color=1, perturb=False, get_normal_image=True, env_net_index=0, material=dict(SPH_CPU_MATERIAL))
    out["miss|rays_d"] = side
    for k, v in r.items():
        out[f"miss|{k}"] = g(v)
    print(f"[golden] sph_cpu miss: keys {sorted(r)}")
    np.savez_compressed(OUT / "sph_cpu.npz", **out)


def main():
    if not REFERENCE.exists():
        raise SystemExit("/root/reference is not present: golden vectors can only be regenerated in the build container")
    torch.manual_seed(0)
    torch.set_num_threads(8)
    install_reference()
    if sys.argv[1:] == ["ide"]:
        golden_ide()
        return
    if sys.argv[1:] == ["variants"]:           # only the NeuS / geometric_init / skip_layers fixture
        golden_variants()
        return
    if sys.argv[1:] == ["sph"]:                # only the env-sphere mode fixture
        golden_sph()
        return
    if sys.argv[1:] == ["torch_only"]:         # only the fixtures from the reference's torch-only code
        golden_torch_only()
        return
    if sys.argv[1:] == ["inorm"]:              # only the instanceNorm feature-activation chain
        golden_instance_norm()
        return
    if sys.argv[1:] == ["indir_aabb"]:         # only the obj_aabb variant of the three-pass frame
        golden_indirect_aabb()
        return
    if sys.argv[1:] == ["network_cpu"]:        # only the plain-torch network fixtures
        golden_network_cpu()
        return
    if sys.argv[1:] == ["sph_cpu"]:            # only the env-sphere render function in plain torch
        golden_sph_cpu()
        return
    if sys.argv[1:] == ["nocuda"]:             # only the torch-only render function's fixtures
        golden_non_cuda_ray()
        golden_torch_only_resample()
        return
    if sys.argv[1:] == ["train"]:              # only the training-branch fixtures
        golden_train("toaster", scenes.toaster_scene())
        golden_train("lego", scenes.lego_scene(seed=8), config=OUT / "lego_like.ini", theta=110.0, phi=-40.0)
        return
    golden_ops()
    golden_rays()
    golden_ide()
    golden_torch_only()
    scene = scenes.toaster_scene()
    model, opt = build_reference_model(scene)
    print("[golden] reference model built:", type(model).__name__, "visual_items", opt.visual_items, "indir_ref", opt.indir_ref)
    golden_shading(model, opt, "toaster")
    golden_shading(model, opt, "toaster_rot", env_rot=0.7)
    golden_frame(model, opt, "toaster_48", 48, 48)
    golden_frame(model, opt, "toaster_rot_40", 40, 40, env_rot=2.1, theta=200.0, phi=-35.0)
    # BASELINE config #4: use_renv + indir_ref (three passes) on a concave shape so that reflected rays
    # hit the object again; sdf / beta chosen so that surfaces are opaque enough for weights_sum > 0.9
    scene4 = scenes.toaster_scene(shape=scenes.torus(), seed=3)
    model4, opt4 = build_reference_model(scene4, extra_argv=["--indir_ref"])
    assert opt4.indir_ref and opt4.use_renv
    golden_frame(model4, opt4, "toaster_indir_40", 40, 40, theta=40.0, phi=-50.0)
    golden_indirect_aabb()
    golden_instance_norm()
    golden_relight()
    golden_background()
    golden_grid()
    golden_demo()
    golden_sph()
    golden_variants()
    # BASELINE configs[1]: no environment network, SH-encoded view direction and normal
    model2, opt2 = build_reference_model(scenes.lego_scene(seed=8), config=OUT / "lego_like.ini")
    golden_frame(model2, opt2, "lego_48", 48, 48, theta=110.0, phi=-40.0)
    golden_shading(model2, opt2, "lego", n=1024)
    golden_train("toaster", scenes.toaster_scene())
    golden_train("lego", scenes.lego_scene(seed=8), config=OUT / "lego_like.ini", theta=110.0, phi=-40.0)
    golden_non_cuda_ray()
    golden_torch_only_resample()
    golden_network_cpu()
    golden_sph_cpu()


if __name__ == "__main__":
    main()
