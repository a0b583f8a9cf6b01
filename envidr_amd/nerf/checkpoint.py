"""Checkpoint reader for the render-state subset.

Mirrors the model half of the reference's `Trainer.load_checkpoint` (nerf/utils.py:1564-1631) and
the `color_mlp_path` transplant of `Trainer.__init__` (nerf/utils.py:509-530): same key names
(`encoder.embeddings`, `sdf_net.N.weight`, `sdf_density.beta`, `env_net.N.*`, `diffuse_net.*`,
`color_net.*`, `renv_net.*`, `density_bitfield`, `aabb_infer`, ...), the `swap_env_path` environment
transplant (with the `split_diffuse_env` rename), the grow-to-model-shape rule for mismatching
tensors, `strict=False` loading, and the `mean_count` / `mean_density` side values.

Also reads the environment files the reference ships under ckpts/env_ckpts/: they were written by
`extract_env_ckpt` (nerf/sph_loader.py:356-378), which names the keys `env_net0.weight` (no dot after
`env_net`), so the reference's own `startswith('env_net.')` filter does not pick them up; both
spellings are accepted here.

After loading, the fused renderer's packed weight blobs are dropped (`invalidate_fused`) so that the
next render repacks from the new parameters.
"""
from __future__ import annotations

import re
from typing import Callable, Mapping, Optional, Union

import torch

StateSource = Union[str, Mapping]

_ENV_NODOT = re.compile(r"^env_net(\d+)\.(weight|bias)$")


def _read(src: StateSource, device) -> Mapping:
    if isinstance(src, (str, bytes)) or hasattr(src, "__fspath__"):
        return torch.load(src, map_location=device)
    return src


def _as_tensor(v, device):
    return v.to(device) if isinstance(v, torch.Tensor) else torch.as_tensor(v, device=device)


def env_state_from(src: StateSource, device="cpu") -> dict:
    """the `env_net.*` entries of an environment checkpoint (full model checkpoint or an
    env_ckpts/env_net_<id>.pth file), keyed the way the model names them"""
    ckpt = _read(src, device)
    state = ckpt["model"] if "model" in ckpt else ckpt
    out = {}
    for k, v in state.items():
        m = _ENV_NODOT.match(k)
        if m:
            out[f"env_net.{m.group(1)}.{m.group(2)}"] = _as_tensor(v, device)
        elif k.startswith("env_net."):
            out[k] = _as_tensor(v, device)
    return out


def load_checkpoint(model, checkpoint: StateSource, swap_env_path: Optional[StateSource] = None,
                    log: Callable[[str], None] = lambda s: None) -> dict:
    """Load a reference checkpoint (path or already-read dict) into `model` (an
    `envidr_amd.nerf.network.NeRFNetwork`).  Returns {'missing_keys', 'unexpected_keys', 'load_renv'}.

    `swap_env_path`: a second checkpoint whose `env_net.*` replaces the scene's (relighting,
    README.md:136-146); with `opt.split_diffuse_env` the scene's own environment moves to
    `diffuse_env_net.*` first, as in the reference."""
    device = next(model.parameters()).device
    ckpt = _read(checkpoint, device)
    if "model" not in ckpt:                                    # bare state_dict (utils.py:1578-1582)
        state = {k: _as_tensor(v, device) for k, v in ckpt.items()}
        res = model.load_state_dict(state, strict=False)
        model.invalidate_fused()
        return {"missing_keys": list(res.missing_keys), "unexpected_keys": list(res.unexpected_keys), "load_renv": False}
    state = {k: _as_tensor(v, device) for k, v in ckpt["model"].items()}

    if swap_env_path is not None and swap_env_path != "" and model.opt.use_env_net:
        env_state = env_state_from(swap_env_path, device)
        for k in [k for k in state if k.startswith("env_net.")]:
            if model.opt.split_diffuse_env:
                state["diffuse_" + k] = state[k]
            del state[k]
        state.update(env_state)

    load_renv = any(k.startswith("renv_net.") for k in state)

    own = model.state_dict()
    for k in list(state):
        if k in own and own[k].shape != state[k].shape:
            # reference behaviour: keep the model's tensor and overwrite its leading rows
            log(f"[WARN] shape mismatch: {k}, {tuple(own[k].shape)} != {tuple(state[k].shape)}; extending")
            grown = own[k].clone()
            grown[: state[k].shape[0]] = state[k]
            state[k] = grown

    res = model.load_state_dict(state, strict=False)
    if res.missing_keys:
        log(f"[WARN] missing keys: {res.missing_keys}")
    if res.unexpected_keys:
        log(f"[WARN] unexpected keys: {res.unexpected_keys}")
    if getattr(model, "cuda_ray", False):
        if "mean_count" in ckpt:
            model.mean_count = ckpt["mean_count"]
        if "mean_density" in ckpt:
            model.mean_density = ckpt["mean_density"]
    model.invalidate_fused()
    return {"missing_keys": list(res.missing_keys), "unexpected_keys": list(res.unexpected_keys), "load_renv": load_renv}


def load_color_mlps(model, color_mlp_path: StateSource, resume_mlps=("specular", "diffuse", "renv"), load_renv: bool = False) -> None:
    """transplant pre-trained rendering MLPs (ckpts/rendering_mlps.pth) into `model`
    (reference nerf/utils.py:509-530: `--color_mlp_path`, `--resume_mlps`)"""
    device = next(model.parameters()).device
    state = _read(color_mlp_path, device)["model"]

    def sub(prefix):
        return {k[len(prefix):]: _as_tensor(v, device) for k, v in state.items() if k.startswith(prefix)}

    if "specular" in resume_mlps:
        model.color_net.load_state_dict(sub("color_net."))
    if "diffuse" in resume_mlps:
        model.diffuse_net.load_state_dict(sub("diffuse_net."))
    if "renv" in resume_mlps and model.opt.use_renv and not load_renv and getattr(model, "renv_net", None) is not None:
        renv = sub("renv_net.")
        if renv:
            model.renv_net.load_state_dict(renv)
    model.invalidate_fused()
