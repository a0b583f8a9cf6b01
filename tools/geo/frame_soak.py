"""Determinism soak of the fp32 frame: the headline frame rendered many times, hinted and un-hinted alternately, with and without the tile
layout, must give the same bits every time.  Run on the GPU box: python tools/geo/frame_soak.py [frames]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda:0")
r = FusedRenderer.from_scene(scenes.toaster_scene(), device=dev)
ro, rd = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(800, 800))
KEYS = ("image", "depth", "weights_sum", "normal_image", "diffuse_image", "specular_image", "roughness_image")
ref, bad = None, 0
for i in range(frames):
    out = r.render_frame(ro, rd, 0.7, use_cost_hint=(i % 3 != 0), image_width=800 if i % 2 else 0)
    cur = torch.cat([out[k].reshape(-1) for k in KEYS])
    if ref is None:
        ref = cur.clone()
    elif not torch.equal(cur, ref):
        bad += 1
        print(f"frame {i}: {int((cur != ref).sum())} values differ, max abs {float((cur - ref).abs().max()):.3e}")
print(f"{frames} frames, {bad} differing")
sys.exit(1 if bad else 0)
