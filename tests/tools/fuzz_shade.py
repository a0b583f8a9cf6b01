"""Randomised run of the shading kernel (envidr_shade_samples, fp32 and the split-precision mode) against the oracle chain on random
normals / view directions / features / roughness, including the awkward values: roughness 0 and 1, grazing and back-facing views,
axis-aligned and polar directions, non-unit inputs, ragged sizes.  Test infrastructure (imports the oracle); run on the GPU box:
    python tests/tools/fuzz_shade.py [iterations] [seed]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer
from oracle.py import render_oracle as ro

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
scene = scenes.toaster_scene()
r = FusedRenderer.from_scene(scene)
opt = ro.RenderOptions(ide_mode="exact")
cuda = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
def unit(v):
    return v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-30)
bad, worst = 0, 0.0
for it in range(iters):
    M = int(rng.choice([1, 5, 63, 64, 65, 127, 1000, 4097]))
    n = unit(rng.normal(size=(M, 3)))
    d = unit(rng.normal(size=(M, 3)))
    k = min(M, 12)
    special = np.array([[0, 0, 1], [0, 0, -1], [1, 0, 0], [0, 1, 0], [0, -1, 0], [-1, 0, 0], [1e-8, 0, 1], [0, 1e-8, -1], [0.6, 0.8, 0], [0.6, 0, 0.8],
                        [1, 1, 1], [-1, -1, -1]], np.float64)
    n[:k] = unit(special[:k]); d[-k:] = unit(special[:k])
    if M > 20:
        d[12:16] = -n[12:16]                      # looking straight at the surface
        d[16:20] = n[16:20]                       # from behind
    gf = unit(rng.normal(size=(M, 12)))
    rough = rng.uniform(0, 1, M)
    rough[: max(1, M // 8)] = rng.choice([0.0, 1.0, 1e-6, 0.02], max(1, M // 8))
    nonunit = rng.random() < 0.2
    if nonunit:
        n *= rng.uniform(0.5, 1.5, (M, 1))          # the kernels do not re-normalise their inputs either
    rot = None if rng.random() < 0.3 else float(rng.uniform(0, 6.28))
    want = ro.shade_surface(scene.mlps, n, d, gf, rough.reshape(-1, 1), opt, rot)
    # (the split-precision mode clamps activations at 60 000 before their fp16 split, mlp_split.hip.h: with |n| = 1.4 the reflected
    #  direction's degree-16 terms grow 200-fold and the hidden layers leave that range -- its domain is unit directions, which is what
    #  the geometry stage hands it; non-unit normals are checked on the fp32 kernel only)
    for mode, tol in (("fp32", 2e-5),) + ((("f16x2", 2e-5),) if not nonunit else ()):
        got = r.shade(cuda(n), cuda(d), cuda(gf), cuda(rough), rot, env_precision=mode)
        torch.cuda.synchronize()
        for key in ("c_diffuse", "c_specular"):
            a, b = got[key].cpu().numpy().astype(np.float64), want[key].astype(np.float64)
            if not np.isfinite(a).all():
                bad += 1; print(f"iteration {it} {mode} {key}: not finite (M={M})"); continue
            err = np.abs(a - b).max()
            worst = max(worst, err)
            if err > tol:
                bad += 1
                i = int(np.abs(a - b).max(axis=1).argmax())
                print(f"iteration {it} {mode} {key}: max abs {err:.2e} at sample {i}: n={n[i]}, d={d[i]}, roughness={rough[i]}, rot={rot}")
print(f"{iters} iterations, {bad} findings; worst max-abs colour difference {worst:.2e}")
sys.exit(1 if bad else 0)
