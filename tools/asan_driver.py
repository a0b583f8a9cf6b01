"""TEST TOOLING: drives the AddressSanitizer build of the library (tools/asan/libenvidr_amd_asan.so) WITHOUT PyTorch -- tools/asan_shim/torch
stands in for the handful of torch names the host side touches, on device memory from the ROCm installation's own HIP runtime -- through
  1. every operator case of tests/cases.py (all 31 operators of include/envidr_amd.h; results checked against the CPU oracle, loosely: this run
     is about memory errors), plus a table scatter big enough for the LDS-range kernel and a compaction across many workgroups;
  2. envidr_shade_samples in the fp32 and both split-precision forms (k_shade_samples, k_env_split2, k_env_split);
  3. a 32x32 frame through the single persistent kernel (envidr_render_rays) and, as far as the stand-in carries it, the geometry pipeline.
Run by tools/gpu_asan.sh with LD_PRELOAD = the sanitizer runtime, LD_LIBRARY_PATH = /opt/rocm/lib, HSA_XNACK=1; the sanitizer's own
reports go to ASAN_OPTIONS=log_path."""
import os
import sys
import traceback
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools" / "asan_shim"))           # `import torch` -> the stand-in
os.environ.setdefault("ENVIDR_AMD_LIB", str(ROOT / "tools" / "asan" / "libenvidr_amd_asan.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402  (the stand-in)

assert "asan_shim" in torch.__file__, torch.__file__
from envidr_amd import _lib, scenes  # noqa: E402
from tests import cases  # noqa: E402
from tests.util import run_op  # noqa: E402

print("library:", _lib.LIB_PATH, "ABI", _lib.load().envidr_abi_version(), flush=True)
done = failed = 0


def f64(a):
    return a.view(np.float16).astype(np.float64) if a.dtype == np.int16 else a.astype(np.float64)


for cid, op, args, tol in cases.all_cases():
    try:
        got = run_op("hip", op, *args)
        if op not in ("march_rays_train",):          # (slots handed out by an atomic counter: compared per ray by the GPU tests)
            want = run_op("oracle", op, *args)
            for k, (a, b) in enumerate(zip(got, want)):
                if a is None:
                    continue
                if a.dtype.kind in "iu" and a.dtype != np.int16:
                    ok = np.array_equal(a, b)
                else:
                    ok = np.linalg.norm(f64(a) - f64(b)) <= 5e-3 * max(np.linalg.norm(f64(b)), 1e-30) + 1e-6
                if not ok:
                    failed += 1
                    print(f"MISMATCH {cid} {op} output {k}", flush=True)
        done += 1
    except Exception as e:      # noqa: BLE001
        failed += 1
        print(f"FAILED {cid} {op}: {e!r}", flush=True)
print(f"operator cases through the sanitizer build: {done} run, {failed} failed / mismatched", flush=True)

# the LDS-range scatter (>= 2^15 points) and the look-back compaction over many workgroups
rng = np.random.default_rng(0)
sc = scenes.toaster_scene()
offsets = np.ascontiguousarray(sc.offsets, np.int32)
S = float(np.log2(sc.per_level_scale))
B = 50_000
x = rng.uniform(0, 1, (B, 3)).astype(np.float32)
x[: B // 2] = x[: B // 2] * 0.05 + 0.4
grad = rng.standard_normal((16, B, 2)).astype(np.float32)
a = run_op("hip", "hash_encode_backward", grad, x, sc.table, offsets, np.zeros_like(sc.table), B, 3, 2, 16, S, 16, 0, None, None)[4]
b = run_op("oracle", "hash_encode_backward", grad, x, sc.table, offsets, np.zeros_like(sc.table), B, 3, 2, 16, S, 16, 0, None, None)[4]
print("LDS-range scatter, 50 000 points: rel-L2 vs oracle", float(np.linalg.norm(a - b) / np.linalg.norm(b)), flush=True)
dy = np.zeros((B, 16 * 3 * 2), np.float32)
out = run_op("hip", "hash_encode_forward", x, sc.table, offsets, np.zeros((16, B, 2), np.float32), B, 3, 2, 16, S, 16, 1, dy)
ggx = rng.standard_normal((B, 3)).astype(np.float32)
g2 = run_op("hip", "hash_encode_second_backward", grad, x, sc.table, offsets, B, 3, 2, 16, S, 16, 1, out[4], ggx, np.zeros_like(grad), np.zeros_like(sc.table))[8]
print("second backward through the LDS-range scatter: finite", bool(np.isfinite(g2).all()), flush=True)
alive = rng.integers(0, 10 ** 6, 200_000).astype(np.int32)
alive[rng.uniform(size=alive.size) < 0.4] = -1
res = run_op("hip", "compact_alive", alive.size, alive, np.full(alive.size, -7, np.int32), np.zeros(1, np.int32))
keep = alive[alive >= 0]
print("compact_alive, 200 000 ids:", "ok" if int(res[2][0]) == keep.size and np.array_equal(res[1][:keep.size], keep) else "WRONG", flush=True)

# shading: fp32 and both split-precision kernels
try:
    from envidr_amd.fused import FusedRenderer
    r = FusedRenderer.from_scene(sc, device="cuda")
    M = 3_000
    n = rng.normal(size=(M, 3)).astype(np.float32); n /= np.linalg.norm(n, axis=1, keepdims=True)
    d = rng.normal(size=(M, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    geo = rng.normal(size=(M, 12)).astype(np.float32); geo /= np.linalg.norm(geo, axis=1, keepdims=True)
    rough = rng.uniform(0, 1, M).astype(np.float32)
    dev_args = [torch.from_numpy(v).to("cuda") for v in (n, d, geo)]
    rough_t = torch.from_numpy(rough).to("cuda")
    cols = {}
    for prec in ("fp32", "f16x2", "f16x2_v1"):
        res = r.shade(*dev_args, rough_t, 0.3, env_precision=prec)
        cols[prec] = np.concatenate([res["c_diffuse"].cpu().numpy(), res["c_specular"].cpu().numpy()], 1)
        print(f"shade[{prec}]: {M} samples, finite {bool(np.isfinite(cols[prec]).all())}", flush=True)
    print("split forms identical:", bool(np.array_equal(cols["f16x2"], cols["f16x2_v1"])), " vs fp32 rel-L2",
          float(np.linalg.norm(cols["f16x2"] - cols["fp32"]) / np.linalg.norm(cols["fp32"])), flush=True)
    ro, rd = (torch.from_numpy(v).to("cuda") for v in scenes.camera_rays(32, 32))
    try:
        img = r.render(ro, rd, 0.2, extras=True, out={})["image"].cpu().numpy()
        print("persistent single-kernel frame 32x32: finite", bool(np.isfinite(img).all()), "mean", float(img.mean()), flush=True)
    except Exception:       # noqa: BLE001
        print("persistent frame: not carried by the torch stand-in:\n" + traceback.format_exc()[-800:], flush=True)
    try:
        img = r.render_frame(ro, rd, 0.2, out={})["image"].cpu().numpy()
        print("geometry pipeline + record shading 32x32: finite", bool(np.isfinite(img).all()), "mean", float(img.mean()), flush=True)
    except Exception:       # noqa: BLE001
        print("geometry pipeline frame: not carried by the torch stand-in:\n" + traceback.format_exc()[-800:], flush=True)
except Exception:       # noqa: BLE001
    failed += 1
    print("shading section failed:\n" + traceback.format_exc()[-1500:], flush=True)
print("ASAN DRIVER " + ("DONE: no failure" if failed == 0 else f"DONE: {failed} failures"), flush=True)
