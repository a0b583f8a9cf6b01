"""per-wave run times of k_shade_samples (variant shade_wavetime of tools/geo/build_variants.py; ENVIDR_AMD_LIB selects it)"""
import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer
dev = torch.device("cuda:0")
ro, rd = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(800, 800))
r = FusedRenderer.from_scene(scenes.toaster_scene(), device=dev)
for i in range(3):
    res = r.render_frame(ro, rd, 0.1 * i, out={}, image_width=800)
torch.cuda.synchronize()
st = r._frame
cap = st["cap"]
raw = st["cd"].view(-1)[3 * (cap - 8192):3 * (cap - 8192) + 1024 * 8].view(torch.int64).cpu().numpy().reshape(1024, 4)
t0, t1, cyc, hw = raw[:, 0], raw[:, 1], raw[:, 2], raw[:, 3]
dur = (t1 - t0) / 100.0          # s_memrealtime: 100 MHz -> microseconds
print("waves", len(dur), "duration us: mean %.1f min %.1f max %.1f  std %.1f" % (dur.mean(), dur.min(), dur.max(), dur.std()))
print("start spread us %.1f, end spread us %.1f, kernel span us %.1f" % ((t0.max() - t0.min()) / 100.0, (t1.max() - t1.min()) / 100.0, (t1.max() - t0.min()) / 100.0))
print("cycles: mean %.0f  -> clock %.3f GHz" % (cyc.mean(), cyc.mean() / dur.mean() / 1e3))
rounds = np.ceil((res["n_records"] - np.arange(1024) * 64) / 65536.0)
for k in np.unique(rounds): print("rounds", int(k), "waves", int((rounds == k).sum()), "mean us %.1f" % dur[rounds == k].mean())
xcc = (np.arange(1024) % 8)
for x in range(8): print("xcd", x, "mean us %.1f max %.1f" % (dur[xcc == x].mean(), dur[xcc == x].max()))
