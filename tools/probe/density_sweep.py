import sys; sys.path.insert(0, "/root/repo")
import torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer
dev = torch.device("cuda:0")
ro, rd = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(800, 800))
for bias in (0.005, 0.03, 0.045, 0.06, 0.08):
    r = FusedRenderer.from_scene(scenes.toaster_scene(sdf_bias=bias), device=dev)
    res = r.render_frame(ro, rd, 0.1, out={}, image_width=800)
    res = r.render_frame(ro, rd, 0.1, out={}, image_width=800)
    print("sdf_bias", bias, "samples/ray %.2f" % (res["n_records"] / 640000), "hit frac %.3f" % float((res["weights_sum"] > 0).float().mean()), "mean ws %.3f" % float(res["weights_sum"].mean()))
