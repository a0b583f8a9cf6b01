"""envidr_amd/nerf/network.py against the REFERENCE's NeRFNetwork on the CPU, no kernel involved: configurations whose encoders are the
identity (tests/golden/torch_like.ini) run in plain torch on both sides.  Fixture tests/golden/network_cpu.npz holds the reference's own
initial weights, its per-sample outputs (density / sdf, geometry feature, normal, roughness, colours) and the gradients of a scalar of them
w.r.t. every parameter and the positions (make_golden.py golden_network_cpu), for the SDF family, the plain-density branch (`use_sdf`
off: trunc_exp density, normals = the negated density gradient; reference network.py:424-429,519), the NeuS section alpha, a skip layer,
the geometric initialisation (weight-normalised layers, Softplus), the tanh / instanceNorm feature activations, a separate roughness layer,
detached and annealed normals, and diffuse only."""
from pathlib import Path

import numpy as np
import pytest
import torch

GOLD = Path(__file__).parent / "golden" / "network_cpu.npz"


# tag -> RenderOptions overrides on top of the torch_like.ini base: the same settings as make_golden.NETWORK_CPU_VARIANTS
NETWORK_VARIANTS = {
    "sdf": {},
    "density": {"use_sdf": False},
    "neus": {"use_neus_sdf": True},
    "skip": {"num_layers": 4, "skip_layers": [2]},
    "geoinit": {"geometric_init": True},
    "tanh_separate_roughness": {"geo_feat_act": "tanh", "ensemble_mlp": False, "learn_indir_blend": False},
    "instance_norm_detached_annealed": {"geo_feat_act": "instanceNorm", "detach_normal": True, "normal_anneal_ratio": 0.5},
    "diffuse_only": {"diffuse_only": True},
    # the toaster.ini structure with the identity in place of the integrated-direction encoding (encoding_ref = frequency, zero frequencies)
    "env": {"use_reflected_dir": True, "use_env_net": True, "diffuse_with_env": True, "wo_viewdir": True, "hidden_dim_env": 24,
            "light_intensity_scale": 1.3, "intensity_scale": 0.9},
    "env_add": {"use_reflected_dir": True, "use_env_net": True, "diffuse_with_env": True, "diffuse_env_fusion": "add", "env_feat_dim": 12,
                "hidden_dim_env": 24, "env_feat_act": "tanh"},
    "env_mul_split": {"use_reflected_dir": True, "use_env_net": True, "diffuse_with_env": True, "diffuse_env_fusion": "mul", "env_feat_dim": 12,
                      "split_diffuse_env": True, "hidden_dim_env": 24, "hidden_dim_env_diffuse": 20, "env_wo_bias": True},
    "env_no_diffuse_env": {"use_reflected_dir": True, "use_env_net": True, "hidden_dim_env": 24, "env_feat_act": "instanceNorm"},
    "renv": {"use_reflected_dir": True, "use_env_net": True, "diffuse_with_env": True, "wo_viewdir": True, "hidden_dim_env": 24, "use_renv": True,
             "indir_roughness_thresh": 0.12},
    "renv_fixed_blend": {"learn_indir_blend": False, "use_reflected_dir": True, "use_env_net": True, "diffuse_with_env": True, "hidden_dim_env": 24,
                         "use_renv": True, "indir_roughness_thresh": 0.12},
}
# how forward_color is driven per variant (make_golden.NETWORK_CPU_CALLS): env rotation, reflected radiance [N, 3 or 4]
NETWORK_CALLS = {"env": {"env_rot": 0.7}, "env_mul_split": {"env_rot": -1.9}, "renv": {"r_images": 4, "env_rot": 0.4}, "renv_fixed_blend": {"r_images": 3}}


def _build(tag, g):
    from envidr_amd.nerf.network import NeRFNetwork
    from envidr_amd.nerf.options import RenderOptions
    base = dict(scale=0.8, cuda_ray=False, hidden_dim=32, hidden_dim_color=32, hidden_dim_diffuse=16, encoding_pos="frequency", multires=0, encoding_dir="frequency", multires_dir=0, wo_viewdir=False,
                normal_with_mlp=True, use_n_dot_viewdir=True, use_reflected_dir=False, use_env_net=False, diffuse_with_env=False, use_renv=False,
                encoding_ref="frequency", multires_refdir=0, hidden_dim_env=128, env_feat_dim=16, env_feat_act="", visual_items=["roughness"])
    opt = RenderOptions(**{**base, **NETWORK_VARIANTS[tag]})
    m = NeRFNetwork(encoding="hashgrid", encoding_dir=opt.encoding_dir, bound=opt.bound, cuda_ray=opt.cuda_ray, density_scale=1, min_near=opt.min_near,
                    density_thresh=opt.density_thresh, bg_radius=opt.bg_radius, use_sdf=opt.use_sdf, hidden_dim=opt.hidden_dim,
                    num_layers=opt.num_layers, num_layers_color=opt.num_layers_color, hidden_dim_color=opt.hidden_dim_color,
                    num_levels=opt.num_levels, geo_feat_dim=opt.geo_feat_dim, opt=opt)
    names = [k[len(tag) + 7:] for k in g.files if k.startswith(f"{tag}|param|")]
    ours = dict(m.named_parameters())
    assert sorted(names) == sorted(ours), (sorted(names), sorted(ours))              # the reference's parameter names, nothing more or less
    missing, unexpected = m.load_state_dict({n: torch.from_numpy(g[f"{tag}|param|{n}"]) for n in names}, strict=False)
    assert not unexpected and all(k in ("aabb_train", "aabb_infer") for k in missing), (missing, unexpected)        # (buffers, not parameters)
    return m.train(), names


@pytest.mark.parametrize("tag", list(NETWORK_VARIANTS))
def test_network_mirror_matches_the_reference_on_the_cpu(tag):
    g = np.load(GOLD)
    model, names = _build(tag, g)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    d = torch.from_numpy(g["d"])
    sdfs, sigmas, geo, normals, _ = model.forward_sigma(x, use_sdf_sigma_grad=True, dirs=d, dists=torch.from_numpy(g["dists"]))
    rough = model.roughness
    call = NETWORK_CALLS.get(tag, {})
    n_enc, w_r, n_dot, n_env = model.get_color_mlp_extra_params(normals, d, rough, call.get("env_rot"))
    ri = torch.from_numpy(g["r_images"][:, :call["r_images"]].copy()) if "r_images" in call else None
    rgb = model.forward_color(geo, d, n_enc, w_r, n_dot, True, n_env_enc=n_env, r_images=ri, roughness=rough)
    assert (sdfs is None) == (tag == "density")
    got = {"sdf": sdfs, "sigma": sigmas, "geo_feat": geo, "normal": normals, "roughness": rough, "rgb": rgb, "c_diffuse": model.c_diffuse,
           "c_specular": model.c_specular}
    for k, v in got.items():
        want = g[f"{tag}|{k}"]
        if want.size == 0:                                                           # (not a tensor on the reference's side: None or a plain number)
            assert v is None or not torch.is_tensor(v), (tag, k)
            continue
        a = v.detach().numpy().reshape(want.shape)
        assert np.abs(a - want).max() <= 2e-6 * max(1.0, float(np.abs(want).max())), (tag, k, float(np.abs(a - want).max()))
    loss = (rgb * torch.from_numpy(g["w_rgb"])).sum() + (sigmas.reshape(-1) * torch.from_numpy(g["w_sigma"])).sum()
    params = dict(model.named_parameters())
    grads = torch.autograd.grad(loss, [x, *[params[n] for n in names]], allow_unused=True)
    for n, gr in zip(["x", *names], grads):
        want = g[f"{tag}|grad|{n}"]
        if want.size == 0:
            assert gr is None, (tag, n)
            continue
        a = gr.detach().numpy()
        scale = max(float(np.abs(want).max()), 1e-6)
        assert np.abs(a - want).max() <= 5e-5 * scale, (tag, n, float(np.abs(a - want).max()), scale)


def test_trunc_exp_is_the_references_function():
    """forward exp(x); backward g exp(clamp(x, -15, 15)) -- also where exp itself has overflowed; twice differentiable"""
    from envidr_amd.nerf.activation import trunc_exp
    x = torch.tensor([-100.0, -20.0, -15.0, -1.0, 0.0, 2.0, 15.0, 20.0, 89.0], requires_grad=True)
    y = trunc_exp(x)
    assert torch.equal(y, torch.exp(x.detach()))
    (gx,) = torch.autograd.grad(y.sum(), x, create_graph=True)
    assert torch.equal(gx.detach(), torch.exp(x.detach().clamp(-15, 15)))
    (ggx,) = torch.autograd.grad(gx.sum(), x)
    inside = (x.detach() > -15) & (x.detach() < 15)
    assert torch.allclose(ggx[inside], torch.exp(x.detach()[inside])) and torch.all(ggx[~inside & (x.detach().abs() > 15)] == 0)
