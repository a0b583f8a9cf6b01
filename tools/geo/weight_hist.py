"""Distribution of the compositing weights of the headline frame's records (how many samples are shaded for a weight that cannot
change any output?).  Run on the GPU box: python tools/geo/weight_hist.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer

dev = torch.device("cuda:0")
for name, scene in (("shell (toaster network)", scenes.toaster_scene()),):
    r = FusedRenderer.from_scene(scene, device=dev)
    ro, rd = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(800, 800))
    out = r.render_frame(ro, rd, 0.1)
    st = r._frames[ro.shape[0]]
    M = int(st["counter"].item())
    w = st["w"][:M].double()
    print(name, "records", M, "sum w", float(w.sum()))
    for thr in (0.0, 1e-12, 1e-9, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3):
        sel = w <= thr
        print(f"  w <= {thr:g}: {int(sel.sum())} records ({100.0 * int(sel.sum()) / M:.2f} %), their weight sum {float(w[sel].sum()):.3e}")
