"""Developer probe: oracle render time vs torch thread count on this host (the cpu_baseline leg of bench.py picks the best of a few)."""
import os, sys, time
sys.path.insert(0, '.')
import torch, numpy as np
from envidr_amd import scenes
from oracle.py import render_oracle as ro
scene = scenes.toaster_scene()
rays_o, rays_d = scenes.camera_rays(64, 64)
print("cpu_count", os.cpu_count(), flush=True)
for th in [8, 16, 32, 64, 128]:
    torch.set_num_threads(th)
    ro.render_rays(scene, rays_o[:128], rays_d[:128], ro.RenderOptions(), None)
    t0 = time.perf_counter(); res = ro.render_rays(scene, rays_o, rays_d, ro.RenderOptions(), None); dt = time.perf_counter() - t0
    print(f"threads {th}: {dt:.2f} s  rays/s {4096/dt:.1f} samples/s {res['n_samples']/dt:.0f}", flush=True)
