"""Developer probe (needs a GPU): time of the pieces of a two-phase frame (geometry-only render with sample export, shade, composite)."""
import sys, time, ctypes
sys.path.insert(0, '.')
import torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer, GeometryExport
r = FusedRenderer.from_scene(scenes.toaster_scene())
ro, rd = (torch.from_numpy(a).cuda() for a in scenes.camera_rays(800, 800))
N = ro.shape[0]
def timeit(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
cost = torch.zeros(N, dtype=torch.int16, device="cuda")
out = {}
print("geometry-only render           %.2f ms" % timeit(lambda: r.render(ro, rd, None, extras=True, geometry_only=True, out=out, ray_cost=cost)))
cap = 9_000_000
counter = torch.zeros(1, dtype=torch.int32, device="cuda")
rec = {"ray": torch.empty(cap, dtype=torch.int32, device="cuda"), "idx": torch.empty(cap, dtype=torch.int32, device="cuda"),
       "w": torch.empty(cap, device="cuda"), "normal": torch.empty(cap, 3, device="cuda"), "geo": torch.empty(cap, 12, device="cuda"),
       "rough": torch.empty(cap, device="cuda")}
ex = GeometryExport(counter.data_ptr(), cap, rec["ray"].data_ptr(), rec["idx"].data_ptr(), rec["w"].data_ptr(), rec["normal"].data_ptr(),
                    rec["geo"].data_ptr(), rec["rough"].data_ptr())
def geo_export():
    counter.zero_()
    r.desc.geometry_export = ctypes.pointer(ex)
    try:
        r.render(ro, rd, None, extras=True, geometry_only=True, out=out, ray_cost=cost)
    finally:
        r.desc.geometry_export = None
print("geometry-only render + export  %.2f ms" % timeit(geo_export))
cache = r.cache_geometry(ro, rd)
cout = {}
print("shade + composite (cached)     %.2f ms" % timeit(lambda: r.render_cached(cache, 0.4, out=cout)))
print("full fused render              %.2f ms" % timeit(lambda: r.render(ro, rd, 0.4, extras=True, out=out, ray_cost=cost)))
tp = {}
evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
r.render_two_phase(ro, rd, 0.4, out=tp, ray_cost=cost)
print("two-phase frame                %.2f ms" % timeit(lambda: r.render_two_phase(ro, rd, 0.4, out=tp, ray_cost=cost)))
r.render_two_phase(ro, rd, 0.4, out=tp, ray_cost=cost, events=evs); torch.cuda.synchronize()
print("geometry %.2f  shade %.2f  composite %.2f ms" % tuple(evs[i].elapsed_time(evs[i + 1]) for i in range(3)), tp["n_records"])
ref = r.render(ro, rd, 0.4, extras=True)
print("identical:", all(torch.equal(tp[k], ref[k]) for k in ("image", "depth", "weights_sum", "normal_image", "diffuse_image", "specular_image", "roughness_image")))
