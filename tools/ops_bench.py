"""Standalone operators of the C ABI at headline-like sizes (640 k rays, ~7.7 M samples, the 48.8 MB hash table): time per call, algorithmic
bytes moved and the fraction of the 8 TB/s HBM peak.  These are the kernels of the operator loop (`fused=False`) and of the training branch;
the frame pipeline does not call them.  Run on the GPU box:  python tools/ops_bench.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from envidr_amd import _lib, scenes

dev = torch.device("cuda:0")
sc = scenes.toaster_scene()
ro_, rd_ = scenes.camera_rays(800, 800)
ro, rd = torch.from_numpy(ro_).to(dev), torch.from_numpy(rd_).to(dev)
N = ro.shape[0]
bitfield = torch.from_numpy(sc.bitfield).to(dev)
table = torch.from_numpy(sc.table).to(dev)
offsets = torch.from_numpy(np.ascontiguousarray(sc.offsets, np.int32)).to(dev)
S = float(np.log2(sc.per_level_scale))
aabb = torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev)
rows = []
# The reference's OWN kernels, compiled by hipcc for this GPU (oracle/_ref/libenvidr_ref_hip.so, test infrastructure: present only where it was
# built from /root/reference): the same call on the same device arrays, timed beside ours -- what recompiling the CUDA kernels for CDNA4 gives.
try:
    from oracle import clib
    REF = clib.ref_hip() if clib.ref_hip_available() else None
except Exception:       # noqa: BLE001
    REF = None
def ref_ms(name, args, reps=3):
    if REF is None or not REF.has(name):
        return None
    # `args` may be a list of argument tuples, one per call (kernels that consume their input, e.g. the compositor's alive list)
    per_call = args if isinstance(args, list) else [args] * (reps + 1)
    conv = [[a.data_ptr() if isinstance(a, torch.Tensor) else a for a in one] for one in per_call]
    REF.call(name, *conv[0]); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        REF.call(name, *conv[1 + i])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def bench(name, fn, bytes_, reps=10, note="", ref=None):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    r = ref_ms(*ref) if ref else None
    rows.append((name, ms, bytes_, note, r))
    print(f"{name:44s} {ms:8.3f} ms   {bytes_ / 1e6:9.1f} MB   {bytes_ / ms / 1e6:8.1f} GB/s  ({100 * bytes_ / ms / 1e6 / 8000:4.1f} % of 8 TB/s, "
          f"{100 * bytes_ / ms / 1e6 / 6300:4.1f} % of the 6.3 TB/s a copy reaches)  "
          + (f"[reference kernel on this GPU: {r:.3f} ms = {r / ms:.1f}x]  " if r else "") + note)

nears, fars = torch.empty(N, device=dev), torch.empty(N, device=dev)
bench("near_far_from_aabb (640 k rays)", lambda: _lib.call("near_far_from_aabb", ro, rd, aabb, N, 0.2, nears, fars), N * 32,
      ref=("near_far_from_aabb", (ro, rd, aabb, N, 0.2, nears, fars)))
alive = torch.arange(N, dtype=torch.int32, device=dev)
rays_t = nears.clone()
for n_step in (1, 8):
    M = N * n_step
    xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
    noises = torch.zeros(N, device=dev)
    def f(n_step=n_step, xyzs=xyzs, dirs=dirs, deltas=deltas):
        _lib.call("march_rays", N, n_step, alive, rays_t, ro, rd, 1.0, 0.0, 1024, 1, 128, bitfield, nears, fars, xyzs, dirs, deltas, noises)
    bench(f"march_rays (640 k rays, n_step {n_step})", f, N * 40 + M * 32, note="writes only for samples found; walks empty cells",
          ref=("march_rays", (N, n_step, alive, rays_t, ro, rd, 1.0, 0.0, 1024, 1, 128, bitfield, nears, fars, xyzs, dirs, deltas, noises)))
M = 7_700_000
x01 = torch.rand(M, 3, device=dev)
out = torch.empty(16, M, 2, device=dev)
dy = torch.empty(M, 16 * 3 * 2, device=dev)
bench("hash_encode_forward (7.7 M, no dy_dx)", lambda: _lib.call("hash_encode_forward", x01, table, offsets, out, M, 3, 2, 16, S, 16, 0, None), M * (1024 + 12 + 128),
      note="random points: no locality between neighbours", ref=("hash_encode_forward", (x01, table, offsets, out, M, 3, 2, 16, S, 16, 0, None)))
bench("hash_encode_forward (7.7 M, + dy_dx)", lambda: _lib.call("hash_encode_forward", x01, table, offsets, out, M, 3, 2, 16, S, 16, 1, dy), M * (1024 + 12 + 128 + 384),
      ref=("hash_encode_forward", (x01, table, offsets, out, M, 3, 2, 16, S, 16, 1, dy)))
grad = torch.randn(16, M, 2, device=dev)
gin = torch.zeros(M, 3, device=dev)
bench("hash_encode_backward (7.7 M, inputs only)", lambda: _lib.call("hash_encode_backward", grad, x01, table, offsets, None, M, 3, 2, 16, S, 16, 1, dy, gin), M * (128 + 384 + 12),
      ref=("hash_encode_backward", (grad, x01, table, offsets, None, M, 3, 2, 16, S, 16, 1, dy, gin)))
gtab = torch.zeros_like(table)
bench("hash_encode_backward (7.7 M, + table scatter)", lambda: _lib.call("hash_encode_backward", grad, x01, table, offsets, gtab, M, 3, 2, 16, S, 16, 1, dy, gin),
      M * (128 + 384 + 12 + 12 + 1024), reps=3, note="8 x 16 fp32 atomics per sample", ref=("hash_encode_backward", (grad, x01, table, offsets, gtab, M, 3, 2, 16, S, 16, 1, dy, gin)))
ggx = torch.randn(M, 3, device=dev); gg = torch.zeros(16, M, 2, device=dev); g2 = torch.zeros_like(table)
bench("hash_encode_second_backward (7.7 M)", lambda: _lib.call("hash_encode_second_backward", grad, x01, table, offsets, M, 3, 2, 16, S, 16, 1, dy, ggx, gg, g2),
      M * (128 + 384 + 12 + 12 + 128 + 1024), reps=3, ref=("hash_encode_second_backward", (grad, x01, table, offsets, M, 3, 2, 16, S, 16, 1, dy, ggx, gg, g2)))
# ---- the same grid operators on the points a FRAME really has (round 5): the marched samples of the benchmark's camera, in the two orders a
# caller feeds them in -- ray-major (march_rays_train: consecutive samples of a ray, the training branch) and sample-major (the operator
# loop's iterations: sample k of every alive ray, image-space neighbours side by side) -- against uniformly random points above
from envidr_amd import raymarching as rm
with torch.no_grad():
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    n_rays = 230_000                                           # ~7.7 M marched samples (33 per ray: marched to the far plane, no termination)
    pick = torch.arange(0, N, N / n_rays, device=dev).long()[:n_rays]
    nr, fr_ = nears[pick].contiguous(), fars[pick].contiguous()
    xyz_r, _, _, rays_r = rm.march_rays_train(ro[pick].contiguous(), rd[pick].contiguous(), 1.0, bitfield, 1, 128, nr, fr_, cnt, 9_000_000, False, 128, False, 0.0, 1024)
    Mr = int(rays_r[:, 2].sum().item())
    xyz_r = xyz_r[:Mr].contiguous()
    x01_ray = ((xyz_r + 1) / 2).contiguous()
    ray_of = torch.repeat_interleave(torch.arange(n_rays, device=dev), rays_r[:, 2].long())
    idx_in_ray = torch.arange(Mr, device=dev) - rays_r[:, 1].long()[ray_of]
    order = torch.argsort(idx_in_ray * n_rays + ray_of)
    x01_smp = x01_ray[order].contiguous()
for label, pts in (("ray-major", x01_ray), ("sample-major", x01_smp)):
    Mp = pts.shape[0]
    out_p = torch.empty(16, Mp, 2, device=dev); dy_p = torch.empty(Mp, 96, device=dev)
    grad_p = torch.randn(16, Mp, 2, device=dev); gin_p = torch.zeros(Mp, 3, device=dev); ggx_p = torch.randn(Mp, 3, device=dev); gg_p = torch.zeros(16, Mp, 2, device=dev)
    bench(f"hash_encode_forward ({Mp / 1e6:.1f} M frame samples, {label}, no dy_dx)",
          lambda: _lib.call("hash_encode_forward", pts, table, offsets, out_p, Mp, 3, 2, 16, S, 16, 0, None), Mp * (1024 + 12 + 128),
          ref=("hash_encode_forward", (pts, table, offsets, out_p, Mp, 3, 2, 16, S, 16, 0, None)))
    bench(f"hash_encode_forward ({Mp / 1e6:.1f} M frame samples, {label}, + dy_dx)",
          lambda: _lib.call("hash_encode_forward", pts, table, offsets, out_p, Mp, 3, 2, 16, S, 16, 1, dy_p), Mp * (1024 + 12 + 128 + 384),
          ref=("hash_encode_forward", (pts, table, offsets, out_p, Mp, 3, 2, 16, S, 16, 1, dy_p)))
    bench(f"hash_encode_backward ({Mp / 1e6:.1f} M frame samples, {label}, + table scatter)",
          lambda: _lib.call("hash_encode_backward", grad_p, pts, table, offsets, gtab, Mp, 3, 2, 16, S, 16, 1, dy_p, gin_p), Mp * (128 + 384 + 12 + 12 + 1024), reps=3,
          ref=("hash_encode_backward", (grad_p, pts, table, offsets, gtab, Mp, 3, 2, 16, S, 16, 1, dy_p, gin_p)))
    bench(f"hash_encode_second_backward ({Mp / 1e6:.1f} M frame samples, {label})",
          lambda: _lib.call("hash_encode_second_backward", grad_p, pts, table, offsets, Mp, 3, 2, 16, S, 16, 1, dy_p, ggx_p, gg_p, g2), Mp * (128 + 384 + 12 + 12 + 128 + 1024), reps=3)
    del out_p, dy_p, grad_p, gin_p, ggx_p, gg_p
# the linear-interpolation grid encoder (gridencoder: not on ENVIDR's configured path, same surface) on the same table shape
from envidr_amd import scenes as _sc
goffs_np, _ = _sc.grid_level_offsets()
goffs = torch.from_numpy(np.ascontiguousarray(goffs_np, np.int32)).to(dev)
gtable = torch.rand(int(goffs_np[-1]), 2, device=dev) * 0.2 - 0.1
gout = torch.empty(16, M, 2, device=dev)
bench("grid_encode_forward (7.7 M, no dy_dx)", lambda: _lib.call("grid_encode_forward", x01, gtable, goffs, gout, M, 3, 2, 16, S, 16, None, 0, 0), M * (1024 + 12 + 128),
      ref=("grid_encode_forward", (x01, gtable, goffs, gout, M, 3, 2, 16, S, 16, None, 0, 0)))
bench("grid_encode_forward (8.2 M frame samples, ray-major, no dy_dx)", lambda: _lib.call("grid_encode_forward", x01_ray, gtable, goffs, gout, x01_ray.shape[0], 3, 2, 16, S, 16, None, 0, 0),
      x01_ray.shape[0] * (1024 + 12 + 128), ref=("grid_encode_forward", (x01_ray, gtable, goffs, gout, x01_ray.shape[0], 3, 2, 16, S, 16, None, 0, 0)))
d = torch.nn.functional.normalize(torch.randn(M, 3, device=dev), dim=-1)
o16 = torch.empty(M, 16, device=dev)
bench("sh_encode_forward (7.7 M, degree 4)", lambda: _lib.call("sh_encode_forward", d, o16, M, 3, 4, None), M * (12 + 64), ref=("sh_encode_forward", (d, o16, M, 3, 4, None)))
o27 = torch.empty(M, 27, device=dev)
bench("freq_encode_forward (7.7 M, degree 4)", lambda: _lib.call("freq_encode_forward", d, M, 3, 4, 27, o27), M * (12 + 108), ref=("freq_encode_forward", (d, M, 3, 4, 27, o27)))
o72 = torch.empty(M, 72, device=dev); rough = torch.rand(M, device=dev)
bench("ide_encode_forward (7.7 M, degree 5)", lambda: _lib.call("ide_encode_forward", d, rough, 0.0, M, 5, o72), M * (16 + 288))
n_step = 8
Mc = N * n_step
sig = torch.rand(Mc, device=dev) * 50; rgb = torch.rand(Mc, 3, device=dev)
al = torch.arange(N, dtype=torch.int32, device=dev); rt = torch.zeros(N, device=dev)
dl = torch.full((Mc, 2), 0.0034, device=dev)
ws, dp, im = torch.zeros(N, device=dev), torch.zeros(N, device=dev), torch.zeros(N, 3, device=dev)
als = [torch.arange(N, dtype=torch.int32, device=dev) for _ in range(12)]      # the compositor tombstones finished rays (-1): a fresh list per call
it = iter(als)
bench("composite_rays (640 k rays x 8 samples)", lambda: _lib.call("composite_rays", N, n_step, 1e-4, 1, 0, next(it), rt, sig, rgb, dl, ws, dp, im),
      N * (n_step * 24 + 8 + 40), ref=("composite_rays", [(N, n_step, 1e-4, 1, 0, torch.arange(N, dtype=torch.int32, device=dev), rt, sig, rgb, dl, ws, dp, im) for _ in range(4)]))
g27 = torch.randn(M, 27, device=dev); gi3 = torch.zeros(M, 3, device=dev)
bench("freq_encode_backward (7.7 M, degree 4)", lambda: _lib.call("freq_encode_backward", g27, o27, M, 3, 4, 27, gi3), M * (108 + 108 + 12),
      ref=("freq_encode_backward", (g27, o27, M, 3, 4, 27, gi3)))
dy48 = torch.empty(M, 48, device=dev)
bench("sh_encode_forward (7.7 M, degree 4, + dy_dx)", lambda: _lib.call("sh_encode_forward", d, o16, M, 3, 4, dy48), M * (12 + 64 + 192),
      ref=("sh_encode_forward", (d, o16, M, 3, 4, dy48)))
g16 = torch.randn(M, 16, device=dev)
bench("sh_encode_backward (7.7 M, degree 4)", lambda: _lib.call("sh_encode_backward", g16, d, M, 3, 4, dy48, gi3), M * (64 + 192 + 24),
      ref=("sh_encode_backward", (g16, d, M, 3, 4, dy48, gi3)))
grid = torch.rand(128 ** 3, device=dev); bits = torch.zeros(128 ** 3 // 8, dtype=torch.uint8, device=dev)
bench("packbits (128^3 cells)", lambda: _lib.call("packbits", grid, 128 ** 3 // 8, 0.01, bits), 128 ** 3 * 4 + 128 ** 3 // 8, note="N counts bytes of the bitfield, like the reference",
      ref=("packbits", (grid, 128 ** 3 // 8, 0.01, bits)))
print()
print("| operator | ms | algorithmic MB | GB/s | of 8 TB/s (HBM roofline) | of the 6.3 TB/s a copy reaches | the reference's kernel, compiled by hipcc, on this GPU |")
print("|---|---|---|---|---|---|---|")
for name, ms, b, note, r in rows:
    print(f"| {name} | {ms:.3f} | {b / 1e6:.0f} | {b / ms / 1e6:.0f} | {100 * b / ms / 1e6 / 8000:.1f} % | {100 * b / ms / 1e6 / 6300:.1f} % | " + (f"{r:.3f} ms ({r / ms:.1f}x) |" if r else "-- |"))
