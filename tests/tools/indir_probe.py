"""Developer probe (needs a GPU): time split of the three-pass indirect render (BASELINE configs[3]) at 800 x 800."""
import sys, time
sys.path.insert(0, '.')
import torch
from envidr_amd import scenes
from envidr_amd.nerf.network import NeRFNetwork
from envidr_amd.nerf.options import toaster_options
opt = toaster_options(indir_ref=True)
m = NeRFNetwork.from_scene(scenes.toaster_scene(shape=scenes.torus(), seed=3), opt)
ro, rd = (torch.from_numpy(a).cuda()[None] for a in scenes.camera_rays(800, 800))
kw = dict(staged=True, bg_color=1, perturb=False, get_normal_image=True, max_steps=opt.max_steps, T_thresh=opt.T_thresh, dt_gamma=opt.dt_gamma)
for _ in range(2):
    m.render(ro, rd, **kw)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    t0 = time.perf_counter(); m.render(ro, rd, **kw); torch.cuda.synchronize(); print("frame ms", (time.perf_counter() - t0) * 1e3)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
