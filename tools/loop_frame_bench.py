"""The reference-shaped OPERATOR LOOP (`fused=False`: march_rays -> encoders / MLPs in torch -> composite_rays, the host loop of
cuda_ray.py:277-346) on an 800x800 frame: what a user gets who only swaps the extension modules.  Run on the GPU box."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from envidr_amd import scenes
from tests.test_dropin_gpu import build_model

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
model, opt = build_model(scenes.toaster_scene())
ro, rd = (torch.from_numpy(a).cuda()[None] for a in scenes.camera_rays(800, 800))
kw = dict(staged=True, bg_color=1, perturb=False, get_normal_image=True, max_steps=opt.max_steps, T_thresh=opt.T_thresh, dt_gamma=opt.dt_gamma)
for fused in (False, True):
    model.render(ro, rd, fused=fused, env_rot_radian=0.3, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(frames):
        model.render(ro, rd, fused=fused, env_rot_radian=0.3 + 0.1 * i, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames
    print(f"fused={fused}: {dt * 1e3:.1f} ms per 800x800 frame ({640000 / dt / 1e6:.2f} M rays/s)")
