"""time of the split-precision environment kernel + fp32 heads on one 800x800 frame's records (run on the GPU box)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer
import numpy as np
from envidr_amd import fused
if "--copies" in sys.argv:          # split_copies4 variant: workgroups stream different copies of the blob
    n = int(sys.argv[sys.argv.index("--copies") + 1])
    orig = fused.pack_env_split
    def tiled(env):
        blob, bias = orig(env)
        return np.concatenate([np.concatenate([blob, np.zeros(256, np.uint16)]) for _ in range(n)]), bias
    fused.pack_env_split = tiled
PREC = "f16x2_v1" if "--v1" in sys.argv else "f16x2"
dev = torch.device("cuda:0")
r = FusedRenderer.from_scene(scenes.toaster_scene(), device=dev)
ro, rd = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(800, 800))
out = {}
for i in range(3):
    r.render_frame(ro, rd, 0.1, out=out, env_precision=PREC)
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(5)]
for i in range(5):
    r.render_frame(ro, rd, 0.1, out=out, events=ev[i], wait=False, env_precision=PREC)
r.check_frames(); torch.cuda.synchronize()
print(f"{PREC}: frame {sum(e[0].elapsed_time(e[3]) for e in ev)/5:.2f} ms, geometry {sum(e[0].elapsed_time(e[1]) for e in ev)/5:.2f} ms; shading (split env + fp32 heads) {sum(e[1].elapsed_time(e[2]) for e in ev)/5:.2f} ms")
