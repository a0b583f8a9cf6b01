"""march_rays_train alone on the three training batches of tools/train_ops_bench.py (ours, and the reference's kernel compiled for this GPU)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from envidr_amd import _lib, raymarching, scenes

dev = torch.device("cuda:0")
sc = scenes.toaster_scene()
bitfield = torch.from_numpy(sc.bitfield).to(dev)
aabb = torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev)
def timeit(fn, reps=20, batches=3):
    """ms per call: the fastest of `batches` event-timed loops (a stall of the shared host inside one loop would otherwise be the figure)"""
    fn(); torch.cuda.synchronize()
    best = float("inf")
    for _ in range(batches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best
try:
    from oracle import clib
    REF = clib.ref_hip() if clib.ref_hip_available() else None
except Exception:       # noqa: BLE001
    REF = None
row = []
for side in (64, 128, 0, 800):
    if side:
        ro_, rd_ = scenes.camera_rays(side, side)
    else:
        ro_, rd_ = scenes.camera_rays(800, 800)
        pick = np.random.default_rng(0).choice(ro_.shape[0], 4096, replace=False)
        ro_, rd_ = ro_[pick], rd_[pick]
    ro, rd = torch.from_numpy(ro_).to(dev), torch.from_numpy(rd_).to(dev)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 1.0, bitfield, 1, 128, nears, fars, force_all_rays=True, align=128)
    M = xyzs.shape[0]
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    N_ = ro.shape[0]
    bufs = (torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev), torch.zeros(N_, 3, dtype=torch.int32, device=dev))
    noise = torch.zeros(N_, device=dev)
    margs = (ro, rd, bitfield, 1.0, 0.0, 1024, 1024, N_, 1, 128, M, nears, fars, *bufs, cnt, noise)
    def ours_march():
        cnt.zero_(); _lib.call("march_rays_train", *margs)
    t_m = timeit(ours_march)
    def ref_march():
        cnt.zero_(); REF.call("march_rays_train", *[a.data_ptr() if isinstance(a, torch.Tensor) else a for a in margs])
    r_m = timeit(ref_march, reps=5) if REF is not None and len(sys.argv) > 1 else float("nan")
    row.append(f"{N_} rays {M} samples (max/ray {int(rays[:, 2].max())}): {t_m:.3f} ms [ref {r_m:.3f}]")
print(" | ".join(row))
