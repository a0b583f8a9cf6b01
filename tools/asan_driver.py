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
NO_DEVICE = ("no ROCm-capable device", "no memory for", "shade_records: null pointer")      # (the last: what a refused geometry pass leaves behind)
refused = 0

if torch.DRY:
    # ENVIDR_ASAN_SHIM_DRY=1, no GPU: a rehearsal of THIS DRIVER and of the stand-in (does every host path up to each launch work on the
    # stand-in's tensors?).  Launches end with "no ROCm-capable device"; they are counted and the call reports success, so the host code after
    # them runs too.  Nothing about the kernels is learnt from such a run.
    class _Fn:
        """an entry point of the library: calls go through `_Rehearsal.call`, argtypes / restype set by the binding code land on the real one"""

        def __init__(self, owner, fn):
            object.__setattr__(self, "_owner", owner)
            object.__setattr__(self, "_fn", fn)

        def __call__(self, *a):
            return self._owner.call(self._fn, *a)

        def __getattr__(self, k):
            return getattr(self._fn, k)

        def __setattr__(self, k, v):
            setattr(self._fn, k, v)

    class _Rehearsal:
        def __init__(self, lib):
            self._lib = lib

        def call(self, fn, *a):
            global refused
            rc = fn(*a)
            if isinstance(rc, int) and rc < 0 and any(s in self._lib.envidr_last_error().decode() for s in NO_DEVICE):
                refused += 1
                return 0
            return rc

        def __getattr__(self, name):
            fn = getattr(self._lib, name)
            if not name.startswith("envidr_") or name in ("envidr_last_error", "envidr_abi_version"):
                return fn
            return _Fn(self, fn)

    _real = _lib.load()
    _proxy = _Rehearsal(_real)
    _lib.load = lambda: _proxy


def f64(a):
    return a.view(np.float16).astype(np.float64) if a.dtype == np.int16 else a.astype(np.float64)


for cid, op, args, tol in cases.all_cases():
    try:
        got = run_op("hip", op, *args)
        if op not in ("march_rays_train",) and not torch.DRY:          # (slots handed out by an atomic counter: compared per ray by the GPU tests)
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

def expect(label, ok, detail=""):
    """a result check: counted as a failure on the GPU, only printed in a rehearsal (where every result is whatever the buffers held)"""
    global failed
    print(f"{label}: {'ok' if ok else 'WRONG'} {detail}".rstrip(), flush=True)
    if not ok and not torch.DRY:
        failed += 1


def section(name, fn):
    """one part of the run; a failure is reported and the next part still runs"""
    global failed
    try:
        fn()
    except Exception as e:      # noqa: BLE001
        failed += 1
        print(f"{name} FAILED:\n" + traceback.format_exc()[-1500:], flush=True)


# the LDS-range scatter (>= 2^15 points) and the look-back compaction over many workgroups
rng = np.random.default_rng(0)
sc = scenes.toaster_scene()
offsets = np.ascontiguousarray(sc.offsets, np.int32)
S = float(np.log2(sc.per_level_scale))
B = 50_000
x = rng.uniform(0, 1, (B, 3)).astype(np.float32)
x[: B // 2] = x[: B // 2] * 0.05 + 0.4
grad = rng.standard_normal((16, B, 2)).astype(np.float32)


def scatter_part():
    a = run_op("hip", "hash_encode_backward", grad, x, sc.table, offsets, np.zeros_like(sc.table), B, 3, 2, 16, S, 16, 0, None, None)[4]
    b = run_op("oracle", "hash_encode_backward", grad, x, sc.table, offsets, np.zeros_like(sc.table), B, 3, 2, 16, S, 16, 0, None, None)[4]
    err = float(np.linalg.norm(a - b) / np.linalg.norm(b))
    expect("LDS-range scatter, 50 000 points vs oracle", err < 1e-5, f"(rel-L2 {err:.1e})")
    dy = np.zeros((B, 16 * 3 * 2), np.float32)
    out = run_op("hip", "hash_encode_forward", x, sc.table, offsets, np.zeros((16, B, 2), np.float32), B, 3, 2, 16, S, 16, 1, dy)
    ggx = rng.standard_normal((B, 3)).astype(np.float32)
    g2 = run_op("hip", "hash_encode_second_backward", grad, x, sc.table, offsets, B, 3, 2, 16, S, 16, 1, out[4], ggx, np.zeros_like(grad), np.zeros_like(sc.table))[-1]
    expect("second backward through the LDS-range scatter finite", bool(np.isfinite(g2).all()))


def compaction_part():
    alive = rng.integers(0, 10 ** 6, 200_000).astype(np.int32)
    alive[rng.uniform(size=alive.size) < 0.4] = -1
    res = run_op("hip", "compact_alive", alive.size, alive, np.full(alive.size, -7, np.int32), np.zeros(1, np.int32))
    keep = alive[alive >= 0]
    expect("compact_alive, 200 000 ids", int(res[2][0]) == keep.size and np.array_equal(res[1][:keep.size], keep))



section("LDS-range scatter", scatter_part)
section("look-back compaction", compaction_part)

# shading: fp32 and both split-precision kernels
def shading_part():
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
        expect(f"shade[{prec}], {M} samples, finite", bool(np.isfinite(cols[prec]).all()))
    err = float(np.linalg.norm(cols["f16x2"] - cols["fp32"]) / np.linalg.norm(cols["fp32"]))
    expect("split forms identical, and close to fp32", bool(np.array_equal(cols["f16x2"], cols["f16x2_v1"])) and err < 1e-5, f"(rel-L2 {err:.1e})")
    ro, rd = (torch.from_numpy(v).to("cuda") for v in scenes.camera_rays(32, 32))
    try:
        img = r.render(ro, rd, 0.2, extras=True, out={})["image"].cpu().numpy()
        expect("persistent single-kernel frame 32x32 finite", bool(np.isfinite(img).all()), f"(mean {float(img.mean()):.4f})")
    except Exception:       # noqa: BLE001
        print("persistent frame: not carried by the torch stand-in:\n" + traceback.format_exc()[-800:], flush=True)
    try:
        img = r.render_frame(ro, rd, 0.2, out={})["image"].cpu().numpy()
        expect("geometry pipeline + record shading 32x32 finite", bool(np.isfinite(img).all()), f"(mean {float(img.mean()):.4f})")
    except Exception:       # noqa: BLE001
        print("geometry pipeline frame: not carried by the torch stand-in:\n" + traceback.format_exc()[-800:], flush=True)


section("shading and frames", shading_part)


def more_kernels_part():
    """the frame pipeline with the split-precision shading behind the shade list, a geometry-only frame re-shaded, the per-sample geometry
    entry points, and the dense-layer operators of the training branch (checked against numpy)"""
    from envidr_amd import fused
    from envidr_amd.fused import FusedRenderer
    r = FusedRenderer.from_scene(sc, device="cuda")
    ro, rd = (torch.from_numpy(v).to("cuda") for v in scenes.camera_rays(48, 48))
    a = r.render_frame(ro, rd, 0.2, out={}, env_precision="fp32")["image"].cpu().numpy()
    for prec in ("f16x2", "f16x2_v1"):
        b = r.render_frame(ro, rd, 0.2, out={}, env_precision=prec)["image"].cpu().numpy()
        err = float(np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-30))
        expect(f"frame 48x48 with {prec} shading vs the fp32 frame", err < 1e-5, f"(rel-L2 {err:.1e})")
    geo = r.render_frame(ro, rd, None, out={}, geometry_only=True, buffers="two_pass")
    again = r.render_frame(ro, rd, 0.2, out={}, reuse_geometry=geo, buffers="two_pass")["image"].cpu().numpy()
    expect("geometry-only frame re-shaded identical to the one-call frame", bool(np.array_equal(a, again)))
    xyz = torch.from_numpy(rng.uniform(-0.6, 0.6, (5_000, 3)).astype(np.float32)).to("cuda")
    dt = torch.from_numpy(np.full(5_000, 0.0034, np.float32)).to("cuda")
    ev = r.geometry_eval(xyz, dt, want=("alpha", "sigma", "normal", "geo_feat", "roughness", "blend"))
    expect("geometry_eval, 5 000 positions, finite", all(bool(np.isfinite(v.cpu().numpy()).all()) for v in ev.values()))
    pr = r.geometry_probe(xyz)
    expect("geometry_probe finite", bool(np.isfinite(pr["features"].cpu().numpy()).all()))
    for M, K, N in ((4_099, 72, 256), (1_000, 256, 256), (777, 160, 12), (5_001, 32, 64)):
        x = rng.standard_normal((M, K)).astype(np.float32)
        W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        xt, Wt, bt = (torch.from_numpy(v).to("cuda") for v in (x, W, bias))
        want = x.astype(np.float64) @ W.astype(np.float64).T + bias
        y = fused.linear_rows(xt, Wt, bt, relu=True).cpu().numpy()
        e1 = float(np.abs(y - np.maximum(want, 0)).max())
        gy = rng.standard_normal((M, N)).astype(np.float32)
        gx = fused.linear_rows(torch.from_numpy(gy).to("cuda"), torch.from_numpy(np.ascontiguousarray(W.T)).to("cuda")).cpu().numpy()
        e2 = float(np.abs(gx - gy.astype(np.float64) @ W.astype(np.float64)).max())
        dW, db = fused.linear_weight_grad(xt, torch.from_numpy(gy).to("cuda"))
        e3 = float(np.abs(dW.cpu().numpy() - gy.astype(np.float64).T @ x.astype(np.float64)).max())
        e4 = float(np.abs(db.cpu().numpy() - gy.astype(np.float64).sum(0)).max())
        expect(f"dense layer {K} -> {N} over {M} rows", max(e1, e2) < 1e-4 and max(e3, e4) < 1e-4 * np.sqrt(M),
               f"(forward {e1:.1e}, input gradient {e2:.1e}, weight gradient {e3:.1e}, bias gradient {e4:.1e})")


section("frame pipeline variants, geometry entry points, dense layers", more_kernels_part)
if torch.DRY:
    print(f"REHEARSAL without a GPU: {refused} launches refused by the runtime and skipped", flush=True)
print("ASAN DRIVER " + ("DONE: no failure" if failed == 0 else f"DONE: {failed} failures"), flush=True)
