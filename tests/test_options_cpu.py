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


@pytest.mark.parametrize("name", ["error_bound_sample", "env_sph_mode", "render_env_on_sphere", "unwrap_env_sphere", "plot_roughness"])
def test_options_outside_the_path_are_refused(name):
    with pytest.raises(NotImplementedError):
        _build(**{name: True})


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
