"""One training step of the toaster network (run_cuda's training branch: 4 096 rays, eikonal loss on) -- wall time per step and, under
rocprofv3 --kernel-trace --stats, where it goes.  Run on the GPU box:  python tools/train_step_bench.py [steps]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from envidr_amd import scenes
from tests.test_dropin_gpu import build_model

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
model, opt = build_model(scenes.toaster_scene())
model.train(); opt.eikonal_loss = True
ro_, rd_ = scenes.camera_rays(800, 800)
pick = np.random.default_rng(0).choice(ro_.shape[0], 4096, replace=False)
ro, rd = torch.from_numpy(ro_[pick]).cuda()[None], torch.from_numpy(rd_[pick]).cuda()[None]
target = torch.rand(1, 4096, 3, device="cuda")
optim = torch.optim.Adam(model.parameters(), lr=1e-4)
def step():
    optim.zero_grad(set_to_none=True)
    res = model.render(ro, rd, staged=False, bg_color=1, perturb=True, force_all_rays=False, max_steps=opt.max_steps, T_thresh=opt.T_thresh, dt_gamma=opt.dt_gamma)
    loss = (res["image"] - target).pow(2).mean() + 0.1 * (res["sdf_gradients"].norm(dim=-1) - 1).pow(2).mean()
    loss.backward()
    optim.step()
    return res
for _ in range(3):
    res = step()
# three timed batches: wall time per step (fastest and slowest batch) and the HOST's share -- how long Python + autograd need to enqueue a step
# (the loop's time before the final synchronize).  ~370 launches a step: where the enqueue time is the wall time the step is host-bound,
# and a loaded or slower host (the GPU boxes share theirs between four jobs) shows up one to one.
walls, hosts = [], []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    walls.append((t2 - t0) / steps); hosts.append((t1 - t0) / steps)
dt = min(walls)
first = (dt, list(walls), list(hosts))
# The figures above are the regime of a run's FIRST 16 steps: mean_count is still 0, so march_rays_train sizes its outputs for max_steps samples
# per ray and reads the sample count back (a host synchronisation per step).  From the first update_extra_state on (every 16 steps, reference
# utils.py train_one_epoch) the marcher allocates mean_count samples and nothing is read back.  The same bookkeeping, without touching the
# synthetic scene's occupancy grid:
n_seen = min(16, int(model.local_step))
if n_seen > 0:
    model.mean_count = int(model.step_counter[:n_seen, 0].sum().item() / n_seen)
    model.local_step = 0
    for _ in range(3):
        res = step()
    walls, hosts = [], []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            res = step()
            if model.local_step >= 16:
                model.local_step = 0            # (update_extra_state resets it every 16 steps)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        walls.append((t2 - t0) / steps); hosts.append((t1 - t0) / steps)
    print(f"steady state (mean_count = {model.mean_count} samples per batch, no read-back): {min(walls) * 1e3:.2f} ms per step; batches of {steps}: "
          f"{', '.join(f'{w * 1e3:.2f}' for w in walls)} ms, host enqueue {', '.join(f'{h * 1e3:.2f}' for h in hosts)} ms")
dt, walls, hosts = first
print(f"training step: 4096 rays, {int(res['sigmas'].shape[0])} samples: {dt * 1e3:.2f} ms per step ({4096 / dt / 1e6:.2f} M rays/s); "
      f"batches of {steps}: {', '.join(f'{w * 1e3:.2f}' for w in walls)} ms, of which the host needs {', '.join(f'{h * 1e3:.2f}' for h in hosts)} ms to enqueue")
