"""per-section clocks of the per-ray rounds of a cold frame (variant `rays_timers` of tools/geo/build_variants.py):
    ENVIDR_AMD_LIB=tools/geo/variants/rays_timers.so python tools/geo/rays_timers_probe.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer
dev = torch.device("cuda:0")
r = FusedRenderer.from_scene(scenes.toaster_scene(), device=dev)
o, d = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(800, 800))
names = ["prologue+count", "composite", "first hit / counting march", "allocate + write march", "state write-back", "outputs"]
for hint in (False, True):
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r.render_frame(o, d, 0.1, geometry_only=True, use_cost_hint=hint)
        e1.record()
    torch.cuda.synchronize()
    print(f"geometry-only frame: {e0.elapsed_time(e1):.3f} ms")
    c = r._frame["ws"][:1024].view(torch.int32).cpu().numpy().astype(np.int64)
    print(f"--- use_cost_hint={hint}  (units: 64 s_memtime ticks ~ 64 x 10 ns if 100 MHz; ratios are what matter)")
    for rnd in range(8):
        w = c[80 + 16 * rnd: 96 + 16 * rnd]
        if w[14] == 0:
            continue
        secs = "  ".join(f"{names[s][:14]}: avg {w[2 * s] / w[14]:7.1f} max {w[2 * s + 1]:6d}" for s in range(6))
        print(f"round {rnd}: blocks {w[14]:5d}  block avg {w[12] / w[14]:8.1f} max {w[13]:7d} | {secs}")
