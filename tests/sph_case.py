"""Shared by the CPU and GPU suites: the env-sphere fixture (tests/golden/sph_render.npz, rendered by the imported reference's
model.render() -> run_sph with the weights it ships; tests/golden/make_golden.py golden_sph) as a scene for the oracle and as a
state_dict for the envidr_amd model."""
from pathlib import Path

import numpy as np

from envidr_amd import scenes

GOLD = Path(__file__).parent / "golden" / "sph_render.npz"


def load():
    return np.load(GOLD)


def scene_from(g) -> scenes.SceneParams:
    """hash table regenerated from the seeded recipe; the shipped networks from the fixture"""
    offsets, pls = scenes.hash_level_offsets()
    pair = lambda prefix, i: (g[f"{prefix}{i}.weight"], g[f"{prefix}{i}.bias"])
    mlps = {"sdf": [pair("sdf/", i) for i in (0, 2, 4)], "env": [pair("env/env_net", i) for i in range(4)],
            "diffuse": [pair("mlps/diffuse_net.", i) for i in range(2)], "specular": [pair("mlps/color_net.", i) for i in range(3)]}
    return scenes.SceneParams(bitfield=np.zeros(128 ** 3 // 8, np.uint8), offsets=offsets, per_level_scale=pls,
                              table=scenes.sphere_table(g["xyz_encoding"]), mlps=mlps, beta=float(g["beta"]))


def material_of(g) -> dict:
    m = g["material"]
    return {"roughness": float(m[0]), "metallic": float(m[1]), "color": [float(v) for v in m[2:5]] + [1.0]}


def rays(g, tag):
    res = int(g[f"{tag}|res"])
    return res, *scenes.camera_rays(res, res, theta=float(g[f"{tag}|theta"]), phi=float(g[f"{tag}|phi"]), radius=4.0, scale=float(g["scale"]))


def build_model(g, device="cuda"):
    """envidr_amd's NeRFNetwork in the env-sphere mode (neural_renderer.ini options), loaded through the state_dict keys a reference
    checkpoint of that mode has: sdf_net.N, env_nets.K.N, diffuse_net.N, color_net.N, encoder.embeddings, sdf_density.beta"""
    import torch
    from envidr_amd.nerf.network import NeRFNetwork
    from envidr_amd.nerf.options import EnvOptions, neural_renderer_options
    opt = neural_renderer_options(env_sph_radius=float(g["radius"]))
    m = NeRFNetwork(encoding="hashgrid", encoding_dir=opt.encoding_dir, bound=opt.bound, cuda_ray=opt.cuda_ray, density_scale=1,
                    min_near=opt.min_near, density_thresh=opt.density_thresh, bg_radius=opt.bg_radius, use_sdf=opt.use_sdf,
                    hidden_dim=opt.hidden_dim, num_layers=opt.num_layers, num_layers_color=opt.num_layers_color,
                    hidden_dim_color=opt.hidden_dim_color, num_levels=opt.num_levels, geo_feat_dim=opt.geo_feat_dim, opt=opt, env_opt=EnvOptions())
    sc = scene_from(g)
    k = int(g["env_net_index"])
    sd = {"encoder.embeddings": torch.from_numpy(sc.table), "sdf_density.beta": torch.tensor(sc.beta)}
    for name, attr in [("sdf", "sdf_net"), ("env", f"env_nets.{k}"), ("diffuse", "diffuse_net"), ("specular", "color_net")]:
        for i, (W, b) in enumerate(sc.mlps[name]):
            sd[f"{attr}.{i}.weight"], sd[f"{attr}.{i}.bias"] = torch.from_numpy(W), torch.from_numpy(b)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    return m.to(device).eval(), opt
