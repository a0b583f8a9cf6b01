"""Race screen for the LDS-DMA weight stream of k_env_split: the split-precision frame rendered many times, at several sizes, must
give the same bits every time (a fragment read before its DMA landed, or overwritten while still being read, shows up as rare
differing tiles).  Run on the GPU box:  python tools/geo/split_soak.py [frames]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer

dev = torch.device("cuda:0")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
r = FusedRenderer.from_scene(scenes.toaster_scene(), device=dev)
bad = 0
for side in (800, 200, 56):
    ro, rd = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(side, side))
    ref = None
    n = frames if side == 800 else frames * 4
    for i in range(n):
        out = r.render_frame(ro, rd, 0.1, env_precision="f16x2")
        img = torch.cat([out["image"].reshape(-1), out["specular_image"].reshape(-1), out["diffuse_image"].reshape(-1)]).clone()
        if ref is None:
            ref = img
        elif not torch.equal(img, ref):
            bad += 1
            print(f"{side}x{side} frame {i}: {int((img != ref).sum())} values differ, max abs {float((img - ref).abs().max()):.3e}")
    f32 = r.render_frame(ro, rd, 0.1)["image"].reshape(-1)
    rel = float(torch.linalg.norm(ref[: f32.numel()] - f32) / torch.linalg.norm(f32))
    print(f"{side}x{side}: {n} split frames, rel-L2 vs the fp32 frame {rel:.2e}")
print("differing frames:", bad)
sys.exit(1 if bad else 0)
