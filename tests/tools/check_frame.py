"""Developer check (needs a GPU): one small frame against the oracle, then 800x800 timings.  Lives under tests/ because it uses
the oracle (test infrastructure); run as `python tests/tools/check_frame.py` from the repository root."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer
from oracle.py import render_oracle as ro
from tests.util import rel_l2
scene = scenes.toaster_scene()
r = FusedRenderer.from_scene(scene)
rays_o, rays_d = scenes.camera_rays(36, 36, theta=75.0, phi=-10.0)
t0=time.time(); want = ro.render_rays(scene, rays_o, rays_d, ro.RenderOptions(ide_mode="exact"), None, force_n_step=1); print("oracle s", time.time()-t0, "samples", want["n_samples"])
res = r.render(torch.from_numpy(rays_o).cuda(), torch.from_numpy(rays_d).cuda(), None, extras=True, stats=True)
torch.cuda.synchronize()
print("stats", res["stats"].tolist())
for k in ["image","depth","weights_sum","diffuse_image","specular_image","roughness_image"]:
    a = res[k].cpu().numpy(); b = want[k].reshape(a.shape)
    print(k, "rel_l2", rel_l2(a,b), "maxabs", np.abs(a-b).max(), "nan", np.isnan(a).sum())
n = res["normal_image"].cpu().numpy(); ws = res["weights_sum"].cpu().numpy()[:,None]
print("normal", rel_l2(n*ws+(1-ws), want["normal_image"]))
# timing at 800x800
ro8, rd8 = scenes.camera_rays(800, 800)
o8, d8 = torch.from_numpy(ro8).cuda(), torch.from_numpy(rd8).cuda()
out = {}
for i in range(3):
    torch.cuda.synchronize(); t0=time.time()
    res = r.render(o8, d8, None, extras=True, stats=True, out=out); torch.cuda.synchronize(); dt=time.time()-t0
    st = res["stats"].tolist()
    print(f"800x800: {dt*1e3:.1f} ms  rays/s {640000/dt:.3e} samples {st[0]} samples/s {st[0]/dt:.3e} rounds {st[1]} util {st[0]/(st[1]*64):.3f}")
