"""timing of whole two-phase frames on the geometry pipeline (run on the GPU box)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer
dev = torch.device("cuda:0")
r = FusedRenderer.from_scene(scenes.toaster_scene(), device=dev)
ro, rd = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(800, 800))
out = {}
for i in range(3):
    res = r.render_frame(ro, rd, 0.1 * i, out=out)
print("samples evaluated", res["n_samples"], "records", res["n_records"])
cold = r.render_frame(ro, rd, 0.2, out=out, use_cost_hint=False)
print("without the hint: samples evaluated", cold["n_samples"], "records", cold["n_records"])
r.render_frame(ro, rd, 0.2, out=out)
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(10)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10):
    r.render_frame(ro, rd, 0.03 * i, out=out, events=ev[i], wait=False)
r.check_frames(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
g, s, c = (float(np.mean([e[j].elapsed_time(e[j + 1]) for e in ev])) for j in range(3))
print(f"frame {dt*1e3:.2f} ms: geometry {g:.2f} shading {s:.2f} composite {c:.2f} -> {640000/dt/1e6:.2f} M rays/s")
# the split-precision shading mode (never the headline): time and difference to the fp32 frame
ref = {k: v.clone() for k, v in r.render_frame(ro, rd, 0.2, out={}).items() if hasattr(v, "clone")}
for i in range(2):
    r.render_frame(ro, rd, 0.2, out=out, env_precision="f16x2")
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10):
    r.render_frame(ro, rd, 0.03 * i, out=out, events=ev[i], wait=False, env_precision="f16x2")
r.check_frames(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
g, s, c = (float(np.mean([e[j].elapsed_time(e[j + 1]) for e in ev])) for j in range(3))
res = r.render_frame(ro, rd, 0.2, out=out, env_precision="f16x2")
err = float(torch.linalg.norm(res["image"] - ref["image"]) / torch.linalg.norm(ref["image"]))
print(f"f16x2 frame {dt*1e3:.2f} ms: geometry {g:.2f} shading {s:.2f} composite {c:.2f} -> {640000/dt/1e6:.2f} M rays/s; rel-L2 vs fp32 frame {err:.2e}")
