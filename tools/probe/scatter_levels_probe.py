"""hash_encode_backward's table scatter on a training batch (64x64 patch, ~146 k samples): time with the first k levels only, k = 1 .. 16."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from envidr_amd import _lib, raymarching, scenes
dev = torch.device("cuda:0")
sc = scenes.toaster_scene()
bitfield = torch.from_numpy(sc.bitfield).to(dev)
table = torch.from_numpy(sc.table).to(dev)
offsets = torch.from_numpy(np.ascontiguousarray(sc.offsets, np.int32)).to(dev)
S = float(np.log2(sc.per_level_scale))
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
ro_, rd_ = scenes.camera_rays(64, 64)
ro, rd = torch.from_numpy(ro_).to(dev), torch.from_numpy(rd_).to(dev)
nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 1.0, bitfield, 1, 128, nears, fars, force_all_rays=True, align=128)
M = xyzs.shape[0]
x01 = ((xyzs + 1) / 2).contiguous()
xr = torch.rand(M, 3, device=dev)
gtab = torch.zeros_like(table)
print("offsets", sc.offsets.tolist())
prev = prev_r = 0.0
for k in range(1, 17):
    grad = torch.randn(k, M, 2, device=dev)
    t = timeit(lambda: _lib.call("hash_encode_backward", grad, x01, table, offsets, gtab, M, 3, 2, k, S, 16, 0, None, None))
    tr = timeit(lambda: _lib.call("hash_encode_backward", grad, xr, table, offsets, gtab, M, 3, 2, k, S, 16, 0, None, None))
    # level k-1 alone: offsets shifted so that the call's level 0 is not the same level -- instead time levels [0, k) and difference
    print(f"levels 0..{k - 1}: {t * 1e3:7.1f} us (+{(t - prev) * 1e3:6.1f})   random points: {tr * 1e3:7.1f} us (+{(tr - prev_r) * 1e3:6.1f})   rows of level {k - 1}: {int(sc.offsets[k] - sc.offsets[k - 1])}")
    prev, prev_r = t, tr
