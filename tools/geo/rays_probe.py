"""time of the first per-ray round (k_geo_rays<true>) alone: geometry_only frames under different library variants"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer
dev = torch.device("cuda:0")
r = FusedRenderer.from_scene(scenes.toaster_scene(), device=dev)
ro, rd = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(800, 800))
out = {}
for hint in (True, False):
    for i in range(3):
        res = r.render_frame(ro, rd, 0.0, out=out, geometry_only=True, use_cost_hint=hint)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(); ev[0].record()
    for i in range(5):
        r.render_frame(ro, rd, 0.0, out=out, geometry_only=True, wait=False, use_cost_hint=hint)
    ev[1].record(); torch.cuda.synchronize()
    print(f"hint={hint}: geometry-only frame {ev[0].elapsed_time(ev[1])/5:.3f} ms, samples {res['n_samples']}")
