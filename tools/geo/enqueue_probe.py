"""host time to enqueue one frame (render_frame(wait=False)) against the device time of the frame"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer, FusedOptions
dev = torch.device("cuda:0")
ro, rd = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(800, 800))
for name, r, rot in (("toaster", FusedRenderer.from_scene(scenes.toaster_scene(), device=dev), 0.1),
                     ("lego (configs[1])", FusedRenderer.from_scene(scenes.lego_scene(), FusedOptions(dir_sh_degree=4), device=dev), None)):
    out = {}
    for i in range(3):
        r.render_frame(ro, rd, rot, out=out)
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for i in range(20):
        a = time.perf_counter()
        r.render_frame(ro, rd, rot, out=out, wait=False)
        host.append(time.perf_counter() - a)
    r.check_frames(); torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / 20
    host.sort()
    print(f"{name}: host enqueue median {host[10]*1e3:.3f} ms (min {host[0]*1e3:.3f}), frame {total*1e3:.2f} ms")
