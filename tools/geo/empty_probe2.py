"""more degenerate inputs: training branch with rays that all miss, indirect rendering with nothing hit, surface shading of zero points"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from envidr_amd import scenes
from tests.test_dropin_gpu import build_model

dev = torch.device("cuda:0")
def attempt(name, fn):
    try:
        r = fn(); torch.cuda.synchronize(); print("ok  ", name, "->", r)
    except Exception as e:
        import traceback; print("FAIL", name, "->", type(e).__name__, str(e)[:200]); traceback.print_exc(limit=4)

miss_o = torch.tensor([[0.0, 0.0, -4.0]] * 128, device=dev); miss_d = torch.tensor([[0.0, 1.0, 0.0]] * 128, device=dev)
ro_, rd_ = scenes.camera_rays(24, 24)
hit_o, hit_d = torch.from_numpy(ro_).to(dev), torch.from_numpy(rd_).to(dev)

# training branch
model, opt = build_model(scenes.toaster_scene())
model.train(); opt.eikonal_loss = True
tk = dict(staged=False, bg_color=1, perturb=False, force_all_rays=False, max_steps=opt.max_steps, T_thresh=opt.T_thresh, dt_gamma=opt.dt_gamma)
def train(o, d):
    model.zero_grad()
    res = model.render(o[None], d[None], **tk)
    loss = res["image"].mean() + (res.get("sdf_gradients", torch.zeros(1, 3, device=dev)).norm(dim=-1) - 1).pow(2).mean()
    loss.backward()
    return tuple(res["image"].shape), int(res["sigmas"].shape[0]), float(loss)
attempt("train step, rays hit", lambda: train(hit_o, hit_d))
attempt("train step, all rays miss", lambda: train(miss_o, miss_d))
attempt("train step, zero rays", lambda: train(miss_o[:0], miss_d[:0]))
attempt("train step with perturb", lambda: (tk.__setitem__("perturb", True), train(hit_o, hit_d), tk.__setitem__("perturb", False))[1])
model.eval()

# indirect rendering, nothing hit / everything hit
tor = scenes.toaster_scene()
m2, opt2 = build_model(tor, indir_ref=True, use_renv=True) if "renv" in tor.mlps else (None, None)
if m2 is None:
    try:
        sc = scenes.torus_scene()
        m2, opt2 = build_model(sc, indir_ref=True, use_renv=True)
    except Exception as e:
        print("no renv scene helper:", e)
if m2 is not None:
    kw = dict(staged=True, bg_color=1, perturb=False, get_normal_image=True, max_steps=opt2.max_steps, T_thresh=opt2.T_thresh, dt_gamma=opt2.dt_gamma)
    attempt("indirect render, rays hit", lambda: tuple(m2.render(hit_o[None], hit_d[None], **kw)["image"].shape))
    attempt("indirect render, all rays miss", lambda: float(m2.render(miss_o[None], miss_d[None], **kw)["weights_sum"].abs().max()))
    attempt("indirect render, zero rays", lambda: tuple(m2.render(miss_o[None, :0], miss_d[None, :0], **kw)["image"].shape))
    attempt("indirect render (operator loop), all rays miss", lambda: float(m2.render(miss_o[None], miss_d[None], fused=False, **kw)["weights_sum"].abs().max()))

# shading of known geometry with zero samples
from envidr_amd.fused import FusedShader
sc = scenes.toaster_scene()
sh = FusedShader({k: sc.mlps[k] for k in ("env", "diffuse", "specular")}, ide_degree=5, diffuse_kappa_inv=0.64)
z3, z12, z1 = torch.empty(0, 3, device=dev), torch.empty(0, 12, device=dev), torch.empty(0, device=dev)
attempt("FusedShader.shade of zero samples", lambda: tuple(sh.shade(z3, z3, z12, z1)["c_diffuse"].shape))
from envidr_amd.nerf.render_func.sph_ray import render_surface
attempt("render_surface, all rays miss the sphere", lambda: float(render_surface(sh, miss_o, miss_d, torch.zeros(12, device=dev), 0.5)["mask"].sum()))
