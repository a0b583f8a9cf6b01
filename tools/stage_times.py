"""Stage times (HIP events inside render_frame) of an 800x800 frame: python tools/stage_times.py [toaster|lego|relight]"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer, FusedOptions
kind = sys.argv[1] if len(sys.argv) > 1 else "toaster"
H = W = 800
if kind == "lego":
    r = FusedRenderer.from_scene(scenes.lego_scene(), FusedOptions(dir_sh_degree=4))
elif kind == "relight":
    r = FusedRenderer.from_scene(scenes.toaster_scene(hidden_env=160, ide_deg=4), FusedOptions(ide_degree=4))
else:
    r = FusedRenderer.from_scene(scenes.toaster_scene())
o, d = (torch.from_numpy(a).cuda() for a in scenes.camera_rays(H, W))
out = {}
for _ in range(3):
    r.render_frame(o, d, 0.1, out=out, image_width=W)
steps = 10
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
for i in range(steps):
    r.render_frame(o, d, 0.1, out=out, events=ev[i], wait=False, image_width=W)
torch.cuda.synchronize()
r.check_frames()
g, s, c = (float(np.mean([e[j].elapsed_time(e[j + 1]) for e in ev])) for j in range(3))
print(f"{kind}: geometry {g:.3f} ms  shading {s:.3f} ms  composite {c:.3f} ms  sum {g + s + c:.3f} ms")
