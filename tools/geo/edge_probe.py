import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer
dev = torch.device("cuda:0")
r = FusedRenderer.from_scene(scenes.toaster_scene(), device=dev)
def rays(n, **kw):
    return tuple(torch.from_numpy(a).to(dev) for a in scenes.camera_rays(n, n, **kw))
ro, rd = rays(64)
full = {k: v.clone() for k, v in r.render_frame(ro, rd, 0.3, image_width=64).items() if hasattr(v, "clone")}
# 1. all rays masked off
m0 = r.render_frame(ro, rd, 0.3, ray_mask=torch.zeros(ro.shape[0], dtype=torch.bool, device=dev))
torch.cuda.synchronize()
print("mask all off:", float(m0["weights_sum"].abs().max()), float((m0["image"] - 1).abs().max()), m0["n_records"])
# 2. half masked == full where on
mask = torch.arange(ro.shape[0], device=dev) % 3 == 0
m1 = r.render_frame(ro, rd, 0.3, ray_mask=mask)
torch.cuda.synchronize()
print("mask third:", bool(torch.equal(m1["image"][mask], full["image"][mask])), float(m1["weights_sum"][~mask].abs().max()))
# 3. image_width that does not divide / not multiple of 8
for w in (0, 60, 63, 128, 32, 8):
    o = r.render_frame(ro, rd, 0.3, image_width=w)
    torch.cuda.synchronize()
    print("image_width", w, bool(torch.equal(o["image"], full["image"])), bool(torch.equal(o["depth"], full["depth"])))
# 4. non-contiguous / double / cpu inputs
try:
    o = r.render_frame(ro.double(), rd.double(), 0.3)
    torch.cuda.synchronize(); print("double rays:", bool(torch.equal(o["image"], full["image"])))
except Exception as e: print("double rays ->", type(e).__name__, str(e)[:100])
o = r.render_frame(ro.t().contiguous().t(), rd.t().contiguous().t(), 0.3)
torch.cuda.synchronize(); print("strided rays:", bool(torch.equal(o["image"], full["image"])))
try:
    o = r.render_frame(ro.cpu(), rd.cpu(), 0.3); print("cpu rays: no error?!", o["image"].device)
except Exception as e: print("cpu rays ->", type(e).__name__, str(e)[:120])
# 5. NaN / inf / zero directions
ro2, rd2 = ro.clone(), rd.clone()
rd2[5] = 0; rd2[6] = float("nan"); ro2[7] = float("inf"); rd2[8, 0] = 0
o = r.render_frame(ro2, rd2, 0.3)
torch.cuda.synchronize()
ok = torch.ones(ro.shape[0], dtype=torch.bool, device=dev); ok[5:9] = False
print("bad rays leave the others alone:", bool(torch.equal(o["image"][ok], full["image"][ok])), "bad rows:", o["image"][5:9].tolist())
# 6. large frame 2048^2 runs and its centre crop rays equal a separate render of those rays
ro3, rd3 = rays(2048)
big = r.render_frame(ro3, rd3, 0.3, image_width=2048)
torch.cuda.synchronize()
idx = torch.arange(2048 * 1000 + 900, 2048 * 1000 + 1100, device=dev)
small = r.render_frame(ro3[idx].contiguous(), rd3[idx].contiguous(), 0.3)
torch.cuda.synchronize()
print("2048^2:", big["n_records"], "records; crop equal:", bool(torch.equal(big["image"][idx], small["image"])), "mem GB", torch.cuda.max_memory_allocated() / 2**30)
