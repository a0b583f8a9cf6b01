"""What ONE rank of an N-rank strong-scaling run does, on the one GPU of a test box: render the interleaved 8x8-pixel tiles
shard 0 of N of the 800x800 headline frame and time it (ideal = full frame / N).  Run on the GPU box:
    python tools/geo/shard_probe.py [N ...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from envidr_amd import parallel, scenes
from envidr_amd.fused import FusedRenderer

dev = torch.device("cuda:0")
r = FusedRenderer.from_scene(scenes.toaster_scene(), device=dev)
ro, rd = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(800, 800))
worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
base = None
for world in worlds:
    idx = parallel.tile_shard(800, 800, 0, world).to(dev)
    o, d = ro[idx].contiguous(), rd[idx].contiguous()
    out = {}
    for _ in range(3):
        r.render_frame(o, d, 0.1, out=out)
    steps = 10
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for i in range(steps):
        r.render_frame(o, d, 0.1, out=out, events=ev[i], wait=False)
    t1.record()
    r.check_frames(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / steps
    geo = sum(e[0].elapsed_time(e[1]) for e in ev) / steps
    sh = sum(e[1].elapsed_time(e[2]) for e in ev) / steps
    if base is None:
        base = ms * world
    print(f"shard 0 of {world}: {idx.numel()} rays, {ms:.3f} ms / frame (geometry {geo:.3f}, shading {sh:.3f}); ideal {base / world:.3f} -> efficiency {base / world / ms:.3f}")
