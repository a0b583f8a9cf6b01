"""Indirect rendering (use_renv + indir_ref: geometry pass -> reflected rays -> main pass with reflected radiance) on random scenes: the
fused three-pass path (device masks, no host sync) against the reference-shaped operator loop.  Run on the GPU box:
    python tools/fuzz_indirect.py [first] [count]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from envidr_amd import scenes
from tests.test_dropin_gpu import build_model
from tests.util import rel_l2

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(900 + seed)
    shape = [scenes.torus(), scenes.shell(0.5, 0.06), scenes.ball(0.45)][seed % 3]
    scene = scenes.toaster_scene(shape=shape, seed=30 + seed, beta=float(rng.uniform(0.01, 0.04)), table_scale=float(rng.uniform(0.05, 0.25)))
    model, opt = build_model(scene, indir_ref=True, indir_roughness_thresh=float(rng.choice([0.06, 0.15, 0.3])),
                             indir_max_steps=int(rng.choice([64, 128, 1024])))
    side = int(rng.choice([24, 40, 56]))
    ro_, rd_ = scenes.camera_rays(side, side, theta=float(rng.uniform(0, 360)), phi=float(rng.uniform(-60, 60)))
    ro, rd = torch.from_numpy(ro_).cuda()[None], torch.from_numpy(rd_).cuda()[None]
    kw = dict(staged=True, bg_color=float(rng.uniform(0, 1)), perturb=False, get_normal_image=True, env_rot_radian=float(rng.uniform(0, 6.28)),
              max_steps=opt.max_steps, T_thresh=opt.T_thresh, dt_gamma=opt.dt_gamma)
    try:
        a = model.render(ro, rd, fused=True, **kw)
        b = model.render(ro, rd, fused=False, **kw)
        torch.cuda.synchronize()
    except Exception as e:      # noqa: BLE001
        bad += 1; print(f"seed {seed}: {type(e).__name__} {str(e)[:200]}"); continue
    n = side * side
    for key in ("image", "depth", "weights_sum", "normal_image"):
        x, y = a[key].cpu().numpy().reshape(n, -1), b[key].cpu().numpy().reshape(n, -1)
        per_ray = np.abs(x - y).max(axis=1)
        flipped = per_ray > 1e-3          # ReLU-kink rays; a reflected ray's gate (weights_sum > 0.9 / > 0.3) can also sit on its threshold
        err = rel_l2(x[~flipped], y[~flipped])
        if flipped.sum() > max(3, 2e-3 * n) or err > 1e-4 or not np.isfinite(x).all():
            bad += 1
            print(f"seed {seed} side {side} {key}: {int(flipped.sum())} rays off by > 1e-3, rel-L2 of the rest {err:.2e}")
print(f"{count} scenes, {bad} findings")
sys.exit(1 if bad else 0)
