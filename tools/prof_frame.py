"""Render a few 800x800 frames of the synthetic toaster scene with the fused kernel (profiling target)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H = int(sys.argv[2]) if len(sys.argv) > 2 else 800
kind = sys.argv[3] if len(sys.argv) > 3 else "toaster"
if kind == "lego":      # BASELINE configs[1]: no environment MLP
    from envidr_amd.fused import FusedOptions
    scene = scenes.lego_scene()
    r = FusedRenderer.from_scene(scene, FusedOptions(dir_sh_degree=4))
else:
    scene = scenes.toaster_scene()
    r = FusedRenderer.from_scene(scene)
ro8, rd8 = scenes.camera_rays(H, H)
o8, d8 = torch.from_numpy(ro8).cuda(), torch.from_numpy(rd8).cuda()
out = {}
cost = torch.zeros(H * H, dtype=torch.int16, device="cuda") if (len(sys.argv) > 4 and sys.argv[4] == "hint") else None
for i in range(n):
    torch.cuda.synchronize(); t0 = time.time()
    res = r.render(o8, d8, None, extras=True, stats=True, out=out, ray_cost=cost); torch.cuda.synchronize(); dt = time.time() - t0
    st = res["stats"].tolist()
    if sum(st[4:]) > 0:
        tot = sum(st[4:]); names = ["march", "hash", "sdf", "geom", "ide", "env", "heads", "comp"]
        print("  sections: " + "  ".join(f"{n} {100*v/tot:.1f}%" for n, v in zip(names, st[4:])) + f"  | cycles/round {tot/st[1]:.0f}")
    print(f"{H}x{H}: {dt*1e3:.1f} ms rays/s {H*H/dt:.3e} samples {st[0]} samples/s {st[0]/dt:.3e} rounds {st[1]} util {st[0]/(st[1]*64):.3f}", flush=True)
