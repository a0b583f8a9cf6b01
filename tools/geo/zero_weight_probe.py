"""how many records of a headline frame carry a compositing weight of exactly zero (they add 0 to every image)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer, FusedOptions
dev = torch.device("cuda:0")
ro, rd = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(800, 800))
for name, r in (("toaster", FusedRenderer.from_scene(scenes.toaster_scene(), device=dev)),
                ("lego", FusedRenderer.from_scene(scenes.lego_scene(), FusedOptions(dir_sh_degree=4), device=dev))):
    res = r.render_frame(ro, rd, 0.1 if name == "toaster" else None)
    M = res["n_records"]
    w = r._frame["w"][:M]
    print(name, "records", M, "w == 0:", int((w == 0).sum()), f"({float((w == 0).float().mean()):.3f})", "w < 1e-8:", f"{float((w < 1e-8).float().mean()):.3f}",
          "w < 1e-6:", f"{float((w < 1e-6).float().mean()):.3f}")
