"""geometry stage of frames of a MOVING camera: the per-ray count hint comes from the previous (different) view"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer
dev = torch.device("cuda:0")
r = FusedRenderer.from_scene(scenes.toaster_scene(), device=dev)
views = [tuple(torch.from_numpy(a).to(dev) for a in scenes.camera_rays(800, 800, theta=60.0 + 1.0 * i, phi=20.0 + 0.5 * i)) for i in range(12)]
out = {}
for step, label in ((1, "1 degree per frame"), (4, "4 degrees per frame")):
    for use_hint in (True, False):
        r.render_frame(*views[0], 0.0, out=out, geometry_only=True)
        ms, evaluated, records = [], [], []
        for i in range(step, 12, step):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            res = r.render_frame(*views[i], 0.0, out=out, geometry_only=True, use_cost_hint=use_hint)
            ev[1].record(); torch.cuda.synchronize()
            ms.append(ev[0].elapsed_time(ev[1])); evaluated.append(res["n_samples"]); records.append(res["n_records"])
        print(f"{label}, hint from the previous view={use_hint}: geometry {np.mean(ms):.2f} ms, evaluated / composited = {np.sum(evaluated) / np.sum(records):.3f}")
