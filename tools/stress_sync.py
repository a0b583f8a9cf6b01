"""TEST TOOLING: the kernels with hand-rolled synchronisation run many times on IDENTICAL inputs, every result compared with the first --
a race that changes a result shows up as a differing bit, a lost wake-up as a hang (run under `timeout`):
  * k_env_split2 (LDS-DMA ring with a counted vmcnt + bare s_barrier, eight waves in lock step) and k_env_split: the shaded colours, bit for bit;
  * envidr_compact_alive (decoupled look-back scan across workgroups): the compacted list and its count, bit for bit;
  * the geometry pipeline's device-driven rounds + record shading (work counters claimed a round ahead): a whole 256x256 frame, bit for bit;
  * the LDS-range table scatter (range ownership, LDS atomics): fp32 sums whose order of additions is not fixed -- compared to 1e-5 of the
    largest entry, and the set of touched rows exactly.
    python tools/stress_sync.py [iterations=1000]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from envidr_amd import _lib, scenes
from envidr_amd.fused import FusedRenderer

ITER = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda:0")
r = FusedRenderer.from_scene(scenes.toaster_scene(), device=dev)
rng = np.random.default_rng(0)
M = 700_001            # 21 rounds per workgroup: the weight ring wraps, work counters run ahead
n = rng.normal(size=(M, 3)).astype(np.float32); n /= np.linalg.norm(n, axis=1, keepdims=True)
d = rng.normal(size=(M, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
geo = rng.normal(size=(M, 12)).astype(np.float32); geo /= np.linalg.norm(geo, axis=1, keepdims=True)
args = [torch.from_numpy(x).to(dev) for x in (n, d, geo)]
rough = torch.from_numpy(rng.uniform(0, 1, M).astype(np.float32)).to(dev)
bad = 0
for prec in ("f16x2", "f16x2_v1", "fp32"):
    t0 = time.time()
    first = None
    for it in range(ITER):
        res = r.shade(*args, rough, 0.3, env_precision=prec)
        cur = torch.cat([res["c_diffuse"], res["c_specular"]], 1).clone()
        if first is None:
            first = cur
        elif not torch.equal(cur, first):
            bad += 1
            print(f"shade[{prec}] iteration {it}: {int((cur != first).sum())} values differ from the first run")
    torch.cuda.synchronize()
    print(f"shade[{prec}]: {ITER} runs on {M} samples, all identical: {bad == 0} ({time.time() - t0:.1f} s)")

alive = rng.integers(0, 10 ** 6, 300_000).astype(np.int32)
alive[rng.uniform(size=alive.size) < 0.4] = -1
a = torch.from_numpy(alive).to(dev)
first = None
t0 = time.time()
for it in range(ITER):
    out, cnt = torch.full_like(a, -7), torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.call("compact_alive", a.numel(), a, out, cnt)
    cur = (out.clone(), cnt.clone())
    if first is None:
        first = cur
        keep = alive[alive >= 0]
        assert int(cnt.item()) == keep.size and np.array_equal(out[:keep.size].cpu().numpy(), keep)
    elif not (torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1])):
        bad += 1
        print(f"compact_alive iteration {it}: differs from the first run")
torch.cuda.synchronize()
print(f"compact_alive: {ITER} runs on {a.numel()} ids, all identical ({time.time() - t0:.1f} s)")

ro, rd = (torch.from_numpy(x).to(dev) for x in scenes.camera_rays(256, 256))
first = None
t0 = time.time()
for it in range(max(ITER // 4, 10)):
    res = r.render_frame(ro, rd, 0.2, out={}, use_cost_hint=bool(it & 1))
    cur = torch.cat([res["image"], res["normal_image"], res["depth"][:, None]], 1).clone()
    if first is None:
        first = cur
    elif not torch.equal(cur, first):
        bad += 1
        print(f"frame iteration {it}: {int((cur != first).sum())} values differ from the first run")
print(f"256x256 frame (cold and hinted alternating): {max(ITER // 4, 10)} runs, all identical ({time.time() - t0:.1f} s)")

sc = scenes.toaster_scene()
table = torch.from_numpy(sc.table).to(dev)
offsets = torch.from_numpy(np.ascontiguousarray(sc.offsets, np.int32)).to(dev)
S = float(np.log2(sc.per_level_scale))
B = 1_000_000
x = torch.from_numpy(rng.uniform(0, 1, (B, 3)).astype(np.float32)).to(dev)
x[: B // 2] = x[: B // 2] * 0.05 + 0.4
grad = torch.from_numpy(rng.standard_normal((16, B, 2)).astype(np.float32)).to(dev)
first = None
t0 = time.time()
worst = 0.0
for it in range(max(ITER // 4, 10)):
    gt = torch.zeros_like(table)
    _lib.call("hash_encode_backward", grad, x, table, offsets, gt, B, 3, 2, 16, S, 16, 0, None, None)
    if first is None:
        first = gt.clone()
        scale = float(first.abs().max())
    else:
        worst = max(worst, float((gt - first).abs().max()) / scale)
        if not torch.equal(gt != 0, first != 0) or worst > 1e-5:
            bad += 1
            print(f"table scatter iteration {it}: touched rows differ or sums beyond 1e-5 ({worst:.2e})")
            break
print(f"LDS-range table scatter: {max(ITER // 4, 10)} runs on {B} points, same rows touched, sums within {worst:.1e} of the largest entry ({time.time() - t0:.1f} s)")
print("STRESS " + ("OK: no differing result" if bad == 0 else f"FAILED: {bad} differing results"))
sys.exit(1 if bad else 0)
