"""time of configs[1] (no-environment family) frames on the geometry pipeline + the split mode's heads kernel"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer, FusedOptions
dev = torch.device("cuda:0")
ro, rd = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(800, 800))
plain = FusedRenderer.from_scene(scenes.lego_scene(), FusedOptions(dir_sh_degree=4), device=dev)
out = {}
for i in range(3):
    plain.render_frame(ro, rd, None, out=out)
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(5)]
for i in range(5):
    plain.render_frame(ro, rd, None, out=out, events=ev[i], wait=False)
plain.check_frames(); torch.cuda.synchronize()
g, s, c = (sum(e[j].elapsed_time(e[j + 1]) for e in ev) / 5 for j in range(3))
print(f"configs[1] frame: geometry {g:.2f} shading {s:.2f} composite {c:.2f} ms = {g+s+c:.2f}")
