"""The relighting fixture (tests/golden/frame_relight_40.npz): README.md:136-146 configuration with the
weights the reference ships under ckpts/.  Helpers shared by the CPU and GPU tests."""
from pathlib import Path

import numpy as np

from envidr_amd import scenes

GOLD = Path(__file__).parent / "golden"
OVERRIDES = dict(sh_degree=4, hidden_dim_env=160, intensity_scale=0.8, roughness_scale=0.8)


def fixture():
    return np.load(GOLD / "frame_relight_40.npz")


def shipped_state(g, group):
    """the shipped checkpoint tensors stored in the fixture, as {'model': {key: array}}"""
    return {"model": {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith(group + "/")}}


def relight_scene(g):
    """seeded geometry + the shipped shading MLPs, as oracle SceneParams"""
    scene = scenes.toaster_scene(hidden_env=160, ide_deg=4, seed=4)
    mlps, env = shipped_state(g, "mlps")["model"], shipped_state(g, "env")["model"]
    for name, prefix in [("diffuse", "diffuse_net"), ("specular", "color_net"), ("renv", "renv_net")]:
        scene.mlps[name] = [(mlps[f"{prefix}.{i}.weight"], mlps[f"{prefix}.{i}.bias"]) for i in range(len(scene.mlps[name]))]
    scene.mlps["env"] = [(env[f"env_net{i}.weight"], env[f"env_net{i}.bias"]) for i in range(4)]
    return scene
