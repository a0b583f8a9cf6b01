import sys, numpy as np, torch
sys.path.insert(0, '.')
from envidr_amd import scenes
from tests.test_dropin_gpu import GOLD, build_model
from tests.test_grid_gpu import GRID_SCENE, GRID_POSES, CpuStream
g = np.load(GOLD / "grid_update.npz")
model, _ = build_model(scenes.toaster_scene(**GRID_SCENE))
model.density_grid.zero_(); model.density_bitfield.zero_(); model.grid_rng = CpuStream()
poses = np.stack([scenes.nerf_matrix_to_ngp(scenes.pose_spherical(th, ph, 4.0), scale=0.65) for th, ph in GRID_POSES])
torch.manual_seed(21)
model.mark_untrained_grid(poses, scenes.intrinsics_for(800, 800))
model.update_extra_state()
grid = model.density_grid.cpu().numpy().reshape(-1)
got, want = grid[::61], g["full1/grid_sample"]
d = np.abs(got - want)
idx = np.argsort(-d)[:15]
print("max", d.max(), "mean", d.mean(), "frac>1e-4rel", (d > 1e-4*np.maximum(1, np.abs(want))).mean())
for i in idx: print(i*61, got[i], want[i])
print(np.histogram(np.log10(d[d>0]), bins=10))
