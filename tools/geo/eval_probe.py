"""GPU probe of envidr_geometry_eval: timing on the samples of one 800x800 headline frame + parity against the
persistent kernel's exported geometry records.  Run on the GPU box:  python tools/geo/eval_probe.py [res]"""
import ctypes, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from envidr_amd import scenes, _lib
from envidr_amd import raymarching as rm
from envidr_amd.fused import FusedRenderer

class SamplesOut(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("alpha", "sigma", "normal", "geo_feat", "roughness", "blend")]

def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 800
    dev = torch.device("cuda:0")
    scene = scenes.toaster_scene()
    r = FusedRenderer.from_scene(scene, device=dev)
    lib = r.lib
    lib.envidr_geometry_eval.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p,
                                         ctypes.POINTER(SamplesOut), ctypes.c_void_p]
    lib.envidr_geometry_eval.restype = ctypes.c_int
    ro, rd = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(res, res))
    N = ro.shape[0]
    # samples of the frame: the standalone marcher, 64 steps per ray at once (nobody terminates early in this scene)
    nears, fars = rm.near_far_from_aabb(ro, rd, torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32, device=dev), 0.2)
    alive = torch.arange(N, dtype=torch.int32, device=dev)
    rays_t = nears.clone()
    S = 80
    xyzs, dirs, deltas = rm.march_rays(N, S, alive, rays_t, ro, rd, 1.0, r.bitfield, 1, 128, nears, fars, 128, False, 0, 1024)
    keep = deltas[:, 0] > 0
    xyz = xyzs[keep].contiguous(); dt = deltas[keep][:, 0].contiguous()
    ray_of = (torch.arange(xyzs.shape[0], device=dev) // S)[keep]
    M = xyz.shape[0]
    out = {k: torch.empty(M, *s, device=dev) for k, s in (("alpha", ()), ("sigma", ()), ("normal", (3,)), ("geo", (12,)), ("rough", ()), ("blend", ()))}
    so = SamplesOut(out["alpha"].data_ptr(), out["sigma"].data_ptr(), out["normal"].data_ptr(), out["geo"].data_ptr(),
                    out["rough"].data_ptr(), out["blend"].data_ptr())
    stream = torch.cuda.current_stream(dev).cuda_stream
    def run():
        rc = lib.envidr_geometry_eval(ctypes.byref(r.desc), xyz.data_ptr(), dt.data_ptr(), M, None, ctypes.byref(so), stream)
        assert rc == 0, lib.envidr_last_error()
    run(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    times = []
    for _ in range(8):
        ev[0].record(); run(); ev[1].record(); torch.cuda.synchronize()
        times.append(ev[0].elapsed_time(ev[1]))
    ms = float(np.median(times))
    # parity: the persistent kernel's records of the same frame (positions differ by ulps: n_step = 1 vs 80 schedule)
    cache = r.cache_geometry(ro, rd)
    # order our samples like the cache: (ray, idx) -- ours are already ray-major in march order
    cnt_ours = torch.bincount(ray_of, minlength=N)
    cnt_ref = (cache.offsets[1:] - cache.offsets[:-1]).long()
    off_ours = torch.cumsum(cnt_ours, 0) - cnt_ours
    idx_in_ray = torch.arange(M, device=dev) - off_ours[ray_of]
    sel = idx_in_ray < cnt_ref[ray_of]          # the samples the renderer composited (rays terminate early)
    ok = int(sel.sum()) == cache.n_samples and bool((cnt_ours >= cnt_ref).all())
    msg = f"M={M} median {ms:.3f} ms (min {min(times):.3f}) -> {M/ms/1e6:.3f} G samples/s, {M*26688/ms/1e9:.1f} TFLOP/s | frame-equivalent {cache.n_samples/ (M/ms) :.3f} ms for {cache.n_samples} samples | match {ok}"
    if ok:
        for name, ref in (("normal", cache.normals), ("geo", cache.geo_feat), ("rough", cache.roughness)):
            e = (out[name][sel] - ref).abs()
            msg += f" | {name} max {e.max().item():.1e} mean {e.mean().item():.1e} >1e-3: {(e > 1e-3).float().mean().item():.1e}"
    print(os.environ.get("ENVIDR_AMD_LIB", "default"), "|", msg)

if __name__ == "__main__":
    main()
