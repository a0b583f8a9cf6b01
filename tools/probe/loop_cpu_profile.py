"""Where the HOST's time of the reference-shaped operator loop goes (the loop is host-bound: tools/loop_frame_bench.py, profiles/r05n/
operator_loop_host_time.txt): cProfile over two 800x800 frames of `fused=False`."""
import sys, cProfile, pstats, io, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from envidr_amd import scenes
from tests.test_dropin_gpu import build_model

model, opt = build_model(scenes.toaster_scene())
ro, rd = (torch.from_numpy(a).cuda()[None] for a in scenes.camera_rays(800, 800))
kw = dict(staged=True, bg_color=1, perturb=False, get_normal_image=True, max_steps=opt.max_steps, T_thresh=opt.T_thresh, dt_gamma=opt.dt_gamma)
for _ in range(2):
    model.render(ro, rd, fused=False, env_rot_radian=0.3, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
model.render(ro, rd, fused=False, env_rot_radian=0.3, **kw)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"one frame: host returns after {(t1 - t0) * 1e3:.1f} ms, GPU done after {(t2 - t0) * 1e3:.1f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(2):
    model.render(ro, rd, fused=False, env_rot_radian=0.3, **kw)
pr.disable(); torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(40); print(buf.getvalue()[:8000])
