import sys, time; sys.path.insert(0, "/root/repo")
import torch
from envidr_amd import scenes
from envidr_amd.fused import FusedOptions, FusedRenderer
dev = torch.device("cuda:0")
sc = scenes.toaster_scene()
ro, rd = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(800, 800))
for rep in range(2):
    for skip in (True, False):
        r = FusedRenderer.from_scene(sc, FusedOptions(skip_zero_weight=skip), device=dev)
        out = {}
        for i in range(3): r.render_frame(ro, rd, 0.1 * i, out=out, wait=False, image_width=800)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        t0 = time.perf_counter()
        for i in range(10): r.render_frame(ro, rd, 0.1 * i, out=out, wait=False, image_width=800, events=ev)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print("skip", skip, "ms/frame %.3f" % (dt * 1e3), "shading %.3f" % ev[1].elapsed_time(ev[2]))
