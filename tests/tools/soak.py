"""Developer soak run (needs a GPU): a 2048 x 2048 frame (4.2 M rays: scratch growth, 32-bit counters, geometry export of
~50 M records) and a 100-frame env-rotation loop with the scheduling hint; checks finiteness, determinism and memory."""
import sys, time, math
sys.path.insert(0, '.')
import torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer
r = FusedRenderer.from_scene(scenes.toaster_scene())
ro, rd = (torch.from_numpy(a).cuda() for a in scenes.camera_rays(2048, 2048))
N = ro.shape[0]
cost = torch.zeros(N, dtype=torch.int16, device="cuda")
t0 = time.time(); a = {k: v.clone() for k, v in r.render(ro, rd, 0.3, extras=True, stats=True, ray_cost=cost).items()}; torch.cuda.synchronize()
print(f"2048^2: {time.time()-t0:.3f} s, samples {int(a['stats'][0])}, rays {int(a['stats'][2])}")
b = r.render(ro, rd, 0.3, extras=True, stats=True, ray_cost=cost); torch.cuda.synchronize()
assert all(torch.equal(a[k], b[k]) for k in ("image", "depth", "weights_sum", "normal_image", "diffuse_image", "specular_image"))
assert torch.isfinite(a["image"]).all() and int(a["stats"][2]) == N
t0 = time.time(); cache = r.cache_geometry(ro, rd); torch.cuda.synchronize()
print(f"geometry cache: {cache.n_samples} records in {time.time()-t0:.2f} s, {torch.cuda.memory_allocated()/2**30:.2f} GiB allocated")
c = r.render_cached(cache, 0.3); torch.cuda.synchronize()
assert torch.equal(c["image"], a["image"])
del cache, c
ro, rd = (torch.from_numpy(x).cuda() for x in scenes.camera_rays(800, 800))
cost = torch.zeros(800 * 800, dtype=torch.int16, device="cuda")
out = {}
mem0 = torch.cuda.memory_allocated()
t0 = time.time()
for i in range(100):
    res = r.render(ro, rd, 2 * math.pi * i / 100, extras=True, out=out, ray_cost=cost)
torch.cuda.synchronize()
print(f"100 frames: {(time.time()-t0)*10:.2f} ms/frame, memory growth {torch.cuda.memory_allocated()-mem0} B")
assert torch.isfinite(res["image"]).all()
print("soak ok")
# two-phase frames: 100 frames, then a different batch size (buffers re-keyed), then back
mem0 = torch.cuda.memory_allocated()
t0 = time.time()
for i in range(100):
    res = r.render_two_phase(ro, rd, 2 * math.pi * i / 100, out=out)
torch.cuda.synchronize()
print(f"100 two-phase frames: {(time.time()-t0)*10:.2f} ms/frame, memory growth {(torch.cuda.memory_allocated()-mem0)/2**20:.0f} MiB (record buffers)")
ref = r.render(ro, rd, 2 * math.pi * 99 / 100, extras=True)
assert torch.equal(res["image"], ref["image"]) and torch.isfinite(res["image"]).all()
small = r.render_two_phase(ro[:70001], rd[:70001], 0.5)
assert torch.equal(small["image"], r.render(ro[:70001], rd[:70001], 0.5, extras=True)["image"])
again = r.render_two_phase(ro, rd, 0.5)
assert torch.equal(again["image"], r.render(ro, rd, 0.5, extras=True)["image"])
print("two-phase soak ok")
