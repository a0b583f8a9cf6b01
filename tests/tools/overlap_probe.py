"""Developer probe (needs a GPU): frames issued on one stream vs alternating on two streams (the next frame's waves start on
the SIMDs the current frame's finished waves have left)."""
import sys, time, math
sys.path.insert(0, '.')
import torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer
r = FusedRenderer.from_scene(scenes.toaster_scene())
ro, rd = (torch.from_numpy(a).cuda() for a in scenes.camera_rays(800, 800))
N = ro.shape[0]
def run(n_streams, frames=24):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    outs = [{} for _ in range(n_streams)]
    costs = [torch.zeros(N, dtype=torch.int16, device="cuda") for _ in range(n_streams)]
    for w in range(2 * n_streams):
        with torch.cuda.stream(streams[w % n_streams]):
            r.render(ro, rd, 0.1 * w, extras=True, out=outs[w % n_streams], ray_cost=costs[w % n_streams])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(frames):
        s = i % n_streams
        with torch.cuda.stream(streams[s]):
            r.render(ro, rd, 2 * math.pi * i / 200, extras=True, out=outs[s], ray_cost=costs[s])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / frames
    return dt, outs
ref = {k: v.clone() for k, v in r.render(ro, rd, 2 * math.pi * 23 / 200, extras=True).items()}
for ns in (1, 2, 3):
    dt, outs = run(ns)
    same = torch.equal(outs[23 % ns]["image"], ref["image"])
    print(f"{ns} stream(s): {dt*1e3:.2f} ms/frame, {N/dt/1e6:.2f} Mrays/s, last frame identical to a serial render: {same}")
