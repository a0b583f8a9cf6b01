"""The option fields that shape the render hot path, with the reference's names and defaults
(nerf/options.py) and a preset for configs/scenes/toaster.ini.  The reference drives everything
from one argparse namespace `opt` that is also splatted into `render(**vars(opt))`; any object with
these attributes works here (`argparse.Namespace`, this dataclass, ...)."""
from __future__ import annotations

from dataclasses import dataclass, field


@dataclass
class RenderOptions:
    # geometry / marching
    bound: float = 1.0
    scale: float = 0.33
    cuda_ray: bool = True
    num_steps: int = 512                 # cuda_ray = False (render_func.run): uniform samples per ray ...
    upsample_steps: int = 0              # ... and importance re-samples per ray (reference nerf/options.py defaults)
    dt_gamma: float = 0.0
    max_steps: int = 1024
    T_thresh: float = 1e-4
    min_near: float = 0.2
    density_thresh: float = 10.0
    bg_radius: float = -1.0
    max_ray_batch_cuda: int = -1
    marching_aabb: list = field(default_factory=list)
    obj_aabb: list | None = None
    # position encoding
    encoding_pos: str = "hashgrid_diff"
    num_levels: int = 16
    level_dim: int = 2
    base_resolution: int = 16
    desired_resolution: int = 2048
    log2_hashmap_size: int = 19
    multires: int = 0
    enabled_levels: int = -1
    # sdf network
    use_sdf: bool = True
    use_neus_sdf: bool = False
    num_layers: int = 3
    hidden_dim: int = 64
    geo_feat_dim: int = 12
    geo_feat_act: str = "unitNorm"
    mlp_bias: bool = True
    ensemble_mlp: bool = True
    use_roughness: bool = True
    learn_indir_blend: bool = True
    skip_layers: list = field(default_factory=list)
    geometric_init: bool = False
    inside_outside: bool = False
    geo_init_bias: float = 1.0
    init_variance: float = 0.3          # NeuS density (use_neus_sdf): inv_s = exp(10 variance)
    neus_n_detach: bool = False
    cos_anneal_ratio: float = 1.0
    init_beta: float = 0.1
    beta_min: float = 0.0005
    beta_max: float = 1.0
    roughness_scale: float = 1.0
    roughness_act_scale: float = 0.2
    default_roughness: float = 0.05
    bypass_roughness: bool = False
    detach_normal: bool = False
    normal_anneal_ratio: float = 1.0
    eikonal_loss: bool = False
    # shading
    encoding_dir: str = "frequency"
    multires_dir: int = 0
    multires_normal: int = 0
    multires_refdir: int = 4
    encoding_ref: str = "integrated_dir"
    sh_degree: int = 5
    sh_degree_diffuse: int = 5
    wo_viewdir: bool = True
    normal_with_mlp: bool = True
    use_reflected_dir: bool = True
    use_n_dot_viewdir: bool = True
    use_env_net: bool = True
    num_layers_env: int = 4
    hidden_dim_env: int = 256
    hidden_dim_env_diffuse: int = 256
    env_feat_dim: int = 12
    env_feat_act: str = "unitNorm"
    env_wo_bias: bool = False
    split_diffuse_env: bool = False
    use_diffuse: bool = True
    diffuse_only: bool = False
    diffuse_with_env: bool = True
    diffuse_env_fusion: str = "concat"
    diffuse_kappa_inv: float = 0.64
    num_layers_diffuse: int = 2
    hidden_dim_diffuse: int = 32
    num_layers_color: int = 3
    hidden_dim_color: int = 64
    color_act: str = "sigmoid"
    light_intensity_scale: float = 1.0
    intensity_scale: float = 1.0
    visual_items: list = field(default_factory=lambda: ["specular", "roughness", "diffuse"])
    # indirect reflection
    use_renv: bool = True
    train_renv: bool = False
    indir_ref: bool = False
    indir_only: bool = False
    indir_max_steps: int = 1024
    indir_early_stop_steps: int = 32
    indir_roughness_thresh: float = 0.1
    grad_rays: bool = False
    # env-sphere mode (configs/neural_renderer.ini: the pre-training of the rendering MLPs on an environment-lit sphere; reference
    # renderer.py:376-377 -> render_func.run_sph).  The reference's parser forces cuda_ray off with it (options.py:325-326).
    env_sph_mode: bool = False
    render_env_on_sphere: bool = False
    env_sph_radius: float = 0.95
    backsdf_loss: bool = False
    # modes that select other render functions in the reference (not on this path)
    unwrap_env_sphere: bool = False
    error_bound_sample: bool = False
    debug: bool = False
    plot_roughness: bool = False
    net_init: str = "xavier_uniform"


@dataclass
class EnvOptions:
    """what the network reads of the env-sphere dataset's options (reference nerf/sph_loader.py:18-47 `config_parser`, passed to
    NeRFNetwork as `env_opt`): which material parameters are concatenated to the SDF network's input (network.py:165-175) and how many
    environments -- one environment MLP each (network.py:290-295) -- the dataset has"""
    vary_roughness: bool = True
    vary_metallic: bool = True
    vary_base_color: bool = True
    env_images_names: list = field(default_factory=lambda: [f"env_{i}" for i in range(11)])   # configs/ktx_images_list.txt: 11 names


def neural_renderer_options(**overrides) -> RenderOptions:
    """configs/neural_renderer.ini resolved against nerf/options.py defaults: the env-sphere mode the shipped rendering MLPs and
    environment MLPs were trained in (IDE degree 4, environment MLPs 38-160-160-160-12, SDF network 37-64-64-14: 32 hash features + 5
    material parameters in, sdf + 12 features + roughness out, no indirect blend)"""
    opt = RenderOptions(scale=0.8, cuda_ray=False, env_sph_mode=True, roughness_act_scale=1.0, sh_degree=4, sh_degree_diffuse=4,
                        hidden_dim_env=160, learn_indir_blend=False, use_renv=False, visual_items=["diffuse", "specular"])
    for k, v in overrides.items():
        if not hasattr(opt, k):
            raise AttributeError(f"unknown render option {k!r}")
        setattr(opt, k, v)
    return opt


def plain_options(**overrides) -> RenderOptions:
    """tests/golden/plain_like.ini resolved against nerf/options.py defaults: the plainest SDF configuration (SH view direction, no normal /
    n.v / reflection inputs, no environment network) -- what the torch-only render function `run` can drive; cuda_ray off"""
    opt = RenderOptions(scale=0.8, cuda_ray=False, encoding_dir="sphere_harmonics", sh_degree=4, wo_viewdir=False, normal_with_mlp=False,
                        use_reflected_dir=False, use_n_dot_viewdir=False, use_env_net=False, diffuse_with_env=False, use_renv=False,
                        visual_items=["roughness"])
    for k, v in overrides.items():
        if not hasattr(opt, k):
            raise AttributeError(f"unknown render option {k!r}")
        setattr(opt, k, v)
    return opt


def toaster_options(**overrides) -> RenderOptions:
    """configs/scenes/toaster.ini resolved against nerf/options.py defaults"""
    opt = RenderOptions(scale=0.65)
    for k, v in overrides.items():
        if not hasattr(opt, k):
            raise AttributeError(f"unknown render option {k!r}")
        setattr(opt, k, v)
    return opt
