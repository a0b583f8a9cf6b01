import sys, numpy as np, torch
sys.path.insert(0, '.')
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer
r = FusedRenderer.from_scene(scenes.toaster_scene())
ro, rd = (torch.from_numpy(a).cuda() for a in scenes.camera_rays(800, 800))
c = r.cache_geometry(ro, rd)
counts = (c.offsets[1:] - c.offsets[:-1]).cpu().numpy().astype(np.int16)
import os
os.makedirs('gpurun_out', exist_ok=True)
np.savez_compressed('gpurun_out/ray_counts.npz', counts=counts)
print(counts.sum(), (counts > 0).sum(), counts.max())
