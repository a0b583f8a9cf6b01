"""empty / degenerate inputs through the Python surface (run on the GPU box)"""
import sys, traceback
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer
from tests.test_dropin_gpu import build_model

dev = torch.device("cuda:0")
def attempt(name, fn):
    try:
        r = fn(); torch.cuda.synchronize(); print("ok  ", name, "->", r)
    except Exception as e:
        print("FAIL", name, "->", type(e).__name__, str(e)[:160])

r = FusedRenderer.from_scene(scenes.toaster_scene(), device=dev)
miss_o = torch.tensor([[0.0, 0.0, -4.0]] * 100, device=dev); miss_d = torch.tensor([[0.0, 1.0, 0.0]] * 100, device=dev)
attempt("cache_geometry of rays that all miss", lambda: (lambda c: (c.n_rays, int(c.offsets[-1])))(r.cache_geometry(miss_o, miss_d)))
def cached_miss():
    c = r.cache_geometry(miss_o, miss_d); o = r.render_cached(c, 0.2); return tuple(o["image"].shape), float(o["weights_sum"].abs().max())
attempt("render_cached of an empty cache", cached_miss)
attempt("geometry_eval of zero points", lambda: {k: tuple(v.shape) for k, v in r.geometry_eval(torch.empty(0, 3, device=dev)).items()})
attempt("render (persistent kernel) of zero rays", lambda: tuple(r.render(miss_o[:0], miss_d[:0], 0.1)["image"].shape))
attempt("render_two_phase of zero rays", lambda: tuple(r.render_two_phase(miss_o[:0], miss_d[:0], 0.1)["image"].shape))
attempt("cache_geometry of zero rays", lambda: r.cache_geometry(miss_o[:0], miss_d[:0]).n_rays)

# modules with empty batches
from envidr_amd.hashencoder import HashEncoder
from envidr_amd.gridencoder import GridEncoder
from envidr_amd.freqencoder import FreqEncoder
from envidr_amd.shencoder import SHEncoder
from envidr_amd.ide_encoder import IntegratedDirEncoder
for name, enc, d in (("HashEncoder", HashEncoder(3).cuda(), 3), ("GridEncoder", GridEncoder(3).cuda(), 3), ("FreqEncoder", FreqEncoder(3, degree=4).cuda(), 3),
                     ("SHEncoder", SHEncoder(3, degree=4).cuda(), 3)):
    def f(enc=enc, d=d):
        x = torch.empty(0, d, device=dev, requires_grad=True); y = enc(x); y.sum().backward(); return tuple(y.shape)
    attempt(name + " on an empty batch (+ backward)", f)
attempt("IntegratedDirEncoder on an empty batch", lambda: tuple(IntegratedDirEncoder(3, 5).cuda()(torch.empty(0, 3, device=dev), torch.empty(0, 1, device=dev)).shape))

# raymarching wrappers with no rays
from envidr_amd import raymarching
aabb = torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev)
attempt("near_far_from_aabb of zero rays", lambda: tuple(t.shape for t in raymarching.near_far_from_aabb(miss_o[:0], miss_d[:0], aabb, 0.2)))

# drop-in model: rays that all miss, zero rays, indirect with nothing hit
model, opt = build_model(scenes.toaster_scene())
kw = dict(staged=True, bg_color=1, perturb=False, get_normal_image=True, max_steps=opt.max_steps, T_thresh=opt.T_thresh, dt_gamma=opt.dt_gamma)
attempt("drop-in render, all rays miss", lambda: float(model.render(miss_o[None], miss_d[None], **kw)["weights_sum"].abs().max()))
attempt("drop-in render, zero rays", lambda: tuple(model.render(miss_o[None, :0], miss_d[None, :0], **kw)["image"].shape))
attempt("drop-in render (operator loop), all rays miss", lambda: float(model.render(miss_o[None], miss_d[None], fused=False, **kw)["weights_sum"].abs().max()))
attempt("drop-in render (operator loop), zero rays", lambda: tuple(model.render(miss_o[None, :0], miss_d[None, :0], fused=False, **kw)["image"].shape))
