"""drop-in surface edge cases: batch prefix B = 2, chunked rendering (max_ray_batch_cuda), tensor / None / float bg_color, flags"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from envidr_amd import scenes
from tests.test_dropin_gpu import build_model

model, opt = build_model(scenes.toaster_scene())
ro_, rd_ = scenes.camera_rays(48, 48, theta=30.0, phi=-20.0)
ro, rd = torch.from_numpy(ro_).cuda()[None], torch.from_numpy(rd_).cuda()[None]
kw = dict(staged=True, perturb=False, get_normal_image=True, max_steps=opt.max_steps, T_thresh=opt.T_thresh, dt_gamma=opt.dt_gamma)
base = model.render(ro, rd, bg_color=1, env_rot_radian=0.4, **kw)
torch.cuda.synchronize()
print({k: tuple(v.shape) for k, v in base.items() if hasattr(v, "shape")})
# B = 2: two copies of the view
ro2, rd2 = ro.repeat(2, 1, 1), rd.repeat(2, 1, 1)
b2 = model.render(ro2, rd2, bg_color=1, env_rot_radian=0.4, **kw)
torch.cuda.synchronize()
print("B=2 shapes", {k: tuple(v.shape) for k, v in b2.items() if hasattr(v, "shape")})
print("B=2 equals B=1 twice:", all(bool(torch.equal(b2[k][i], base[k][0])) for k in ("image", "depth", "weights_sum", "normal_image") for i in (0, 1)))
# chunked
model.opt.max_ray_batch_cuda = 1000
ch = model.render(ro, rd, bg_color=1, env_rot_radian=0.4, **kw)
torch.cuda.synchronize()
model.opt.max_ray_batch_cuda = None
print("chunked equals whole:", {k: bool(torch.equal(ch[k], base[k])) for k in base if hasattr(base[k], "shape")})
# bg colours
for bg in (None, 0, 0.5, torch.tensor([0.2, 0.4, 0.6], device="cuda"), torch.rand(48 * 48, 3, device="cuda")):
    o = model.render(ro, rd, bg_color=bg, env_rot_radian=0.4, **kw)
    torch.cuda.synchronize()
    ws = base["weights_sum"][0][:, None]
    bgv = 1.0 if bg is None else bg
    want = base["image"][0] - (1 - ws) * 1.0 + (1 - ws) * bgv
    print("bg", type(bg).__name__, "max |diff| vs re-blended base:", float((o["image"][0] - want).abs().max()))
# flags
for flags in (dict(get_normal_image=False), dict(use_specular_color=False), dict(env_rot_radian=None)):
    k2 = dict(kw); k2.update(flags)
    er = k2.pop("env_rot_radian", 0.4)
    o = model.render(ro, rd, bg_color=1, env_rot_radian=er, **k2)
    torch.cuda.synchronize()
    print(flags, {k: tuple(v.shape) for k, v in o.items() if hasattr(v, "shape")}, "finite:", bool(torch.isfinite(o["image"]).all()))
