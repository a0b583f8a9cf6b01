"""Would consecutive frames of a video overlap if they were enqueued on alternating streams?  Two renderers (own buffers) on two
streams against one renderer on one stream, same frames.  Run on the GPU box: python tools/geo/two_stream_probe.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer

dev = torch.device("cuda:0")
scene = scenes.toaster_scene()
rs = [FusedRenderer.from_scene(scene, device=dev) for _ in range(3)]
ro, rd = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(800, 800))
outs = [dict() for _ in range(3)]
for r, o in zip(rs, outs):
    for _ in range(3):
        r.render_frame(ro, rd, 0.1, out=o, image_width=800)
torch.cuda.synchronize()
steps = 24
for nstreams in (1, 2, 3, 1, 2):
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        k = i % nstreams
        with torch.cuda.stream(streams[k]):
            rs[k].render_frame(ro, rd, 0.1 + 0.01 * i, out=outs[k], wait=False, image_width=800)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for r in rs:
        r.check_frames()
    print(f"{nstreams} stream(s): {1e3 * dt / steps:.3f} ms per frame")
