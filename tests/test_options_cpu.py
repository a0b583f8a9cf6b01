"""Options of the reference that the render path does not implement are refused at construction, never ignored; obj_aabb is
registered the way the reference registers it (nerf/renderer.py:91-97)."""
import numpy as np
import pytest


def _build(**overrides):
    from envidr_amd.nerf.network import NeRFNetwork
    from envidr_amd.nerf.options import toaster_options
    opt = toaster_options(**overrides)
    return NeRFNetwork(encoding="hashgrid", encoding_dir=opt.encoding_dir, bound=opt.bound, cuda_ray=opt.cuda_ray, density_scale=1,
                       min_near=opt.min_near, density_thresh=opt.density_thresh, bg_radius=opt.bg_radius, use_sdf=opt.use_sdf,
                       hidden_dim=opt.hidden_dim, num_layers=opt.num_layers, num_layers_color=opt.num_layers_color,
                       hidden_dim_color=opt.hidden_dim_color, num_levels=opt.num_levels, geo_feat_dim=opt.geo_feat_dim, opt=opt), opt


def test_obj_aabb_is_scaled_clamped_and_not_persistent():
    m, opt = _build(obj_aabb=[-1.5, -1.5, -0.12, 0.35, 1.5, 3.0])
    want = np.clip(np.array([-1.5, -1.5, -0.12, 0.35, 1.5, 3.0], np.float32) * np.float32(opt.scale), -opt.bound, opt.bound)
    assert np.allclose(m.obj_aabb.numpy(), want) and m.obj_aabb[5] == opt.bound
    assert "obj_aabb" not in m.state_dict()                        # register_buffer(..., persistent=False) in the reference
    assert _build()[0].obj_aabb is None and _build(obj_aabb=[])[0].obj_aabb is None
    with pytest.raises(ValueError):
        _build(obj_aabb=[0.0, 1.0, 2.0])


@pytest.mark.parametrize("name", ["error_bound_sample", "unwrap_env_sphere", "plot_roughness"])
def test_options_outside_the_path_are_refused(name):
    with pytest.raises(NotImplementedError):
        _build(**{name: True})


def test_env_sphere_mode_needs_its_dataset_options():
    """env_sph_mode builds one environment MLP per environment and a material-conditioned SDF input: both come from env_opt"""
    from envidr_amd.nerf.network import NeRFNetwork
    from envidr_amd.nerf.options import EnvOptions, neural_renderer_options
    with pytest.raises(ValueError):
        _build(env_sph_mode=True)
    opt = neural_renderer_options()
    m = NeRFNetwork(encoding="hashgrid", encoding_dir=opt.encoding_dir, bound=opt.bound, cuda_ray=opt.cuda_ray, density_scale=1, min_near=opt.min_near,
                    density_thresh=opt.density_thresh, bg_radius=opt.bg_radius, use_sdf=opt.use_sdf, hidden_dim=opt.hidden_dim, num_layers=opt.num_layers,
                    num_layers_color=opt.num_layers_color, hidden_dim_color=opt.hidden_dim_color, num_levels=opt.num_levels,
                    geo_feat_dim=opt.geo_feat_dim, opt=opt, env_opt=EnvOptions(vary_base_color=False))
    assert m.embed_dim == 2 and tuple(m.sdf_net[0].weight.shape) == (64, 34) and tuple(m.sdf_net[2].weight.shape) == (14, 64)
    assert m.env_net is None and len(m.env_nets) == 11 and not m.cuda_ray
    assert m.material_vector({"roughness": 0.3, "metallic": 0.1, "color": [1, 2, 3]}) == [0.3, 0.1]
    keys = set(m.state_dict())
    assert {"env_nets.10.3.weight", "sdf_net.0.weight", "diffuse_net.1.bias", "color_net.2.weight"} <= keys and "density_bitfield" not in keys


def test_sdf_network_variants_have_the_reference_parameter_names():
    """skip_layers narrows the layer in front of the skip; geometric_init weight-normalises (weight_g / weight_v) and forces biases on"""
    m, _ = _build(skip_layers=[1])
    assert [tuple(l.weight.shape) for l in m.sdf_net] == [(32, 32), (64, 64), (15, 64)]
    m, _ = _build(geometric_init=True, mlp_bias=False, use_neus_sdf=True)
    keys = set(m.state_dict())
    assert {"sdf_net.0.weight_g", "sdf_net.0.weight_v", "sdf_net.2.bias", "sdf_density.variance"} <= keys and "sdf_density.beta" not in keys
    assert not m.supports_fused()


def test_every_option_field_is_read_or_refused():
    """no field of RenderOptions may be silently dropped: each one is referenced by the package's render path (or refused above)"""
    import dataclasses
    import re
    from pathlib import Path
    from envidr_amd.nerf.options import RenderOptions
    root = Path(__file__).resolve().parents[1] / "envidr_amd"
    text = "\n".join(p.read_text() for p in [*root.glob("nerf/*.py"), *root.glob("nerf/render_func/*.py"), root / "fused.py", root / "encoding.py"]
                     if p.name != "options.py")
    unread = [f.name for f in dataclasses.fields(RenderOptions) if not re.search(rf"\b{f.name}\b", text)]
    assert unread == [], unread
