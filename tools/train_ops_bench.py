"""The hash encoder's backward kernels on the samples a TRAINING batch actually has (march_rays_train along 4 096 / 16 384 camera rays of the
benchmark scene: consecutive samples of a ray, neighbouring rays of a patch) instead of uniformly random points.  Run on the GPU box."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
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
# the reference's own kernels compiled by hipcc for this GPU (oracle/_ref/libenvidr_ref_hip.so: test infrastructure, present where built)
try:
    from oracle import clib
    REF = clib.ref_hip() if clib.ref_hip_available() else None
except Exception:       # noqa: BLE001
    REF = None
def ref_time(name, *args):
    if REF is None:
        return float("nan")
    conv = [a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args]
    return timeit(lambda: REF.call(name, *conv), reps=5)
for side, what in ((64, "a 64x64 patch"), (128, "a 128x128 patch"), (0, "4096 random pixels of the 800x800 frame")):
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
    x01 = ((xyzs + 1) / 2).contiguous()
    out = torch.empty(16, M, 2, device=dev); dy = torch.empty(M, 96, device=dev)
    grad = torch.randn(16, M, 2, device=dev); gin = torch.zeros(M, 3, device=dev); gtab = torch.zeros_like(table)
    ggx = torch.randn(M, 3, device=dev); gg = torch.zeros(16, M, 2, device=dev); g2 = torch.zeros_like(table)
    t_f = timeit(lambda: _lib.call("hash_encode_forward", x01, table, offsets, out, M, 3, 2, 16, S, 16, 1, dy))
    t_b = timeit(lambda: _lib.call("hash_encode_backward", grad, x01, table, offsets, gtab, M, 3, 2, 16, S, 16, 1, dy, gin))
    t_2 = timeit(lambda: _lib.call("hash_encode_second_backward", grad, x01, table, offsets, M, 3, 2, 16, S, 16, 1, dy, ggx, gg, g2))
    xr = torch.rand(M, 3, device=dev)
    t_br = timeit(lambda: _lib.call("hash_encode_backward", grad, xr, table, offsets, gtab, M, 3, 2, 16, S, 16, 1, dy, gin))
    r_f = ref_time("hash_encode_forward", x01, table, offsets, out, M, 3, 2, 16, S, 16, 1, dy)
    r_b = ref_time("hash_encode_backward", grad, x01, table, offsets, gtab, M, 3, 2, 16, S, 16, 1, dy, gin)
    r_2 = ref_time("hash_encode_second_backward", grad, x01, table, offsets, M, 3, 2, 16, S, 16, 1, dy, ggx, gg, g2)
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
    r_m = timeit(ref_march, reps=5) if REF is not None else float("nan")
    print(f"{what}: {ro.shape[0]} rays, {M} samples: forward+dy_dx {t_f:.3f} ms, backward with table scatter {t_b:.3f} ms "
          f"({M * 256 / t_b / 1e6:.1f} G atomics/s; the same count of random points: {t_br:.3f} ms), second backward {t_2:.3f} ms, march_rays_train {t_m:.3f} ms"
          f"  [the reference's kernels on this GPU: {r_f:.3f} / {r_b:.3f} / {r_2:.3f} / {r_m:.3f} ms]")
