"""Simulation of march-chunk schedules of the geometry pipeline (csrc/geometry_pass.hip, "chunk prediction") on the CPU oracle:
how many samples are evaluated per composited sample, and in how many rounds, for a frame WITHOUT a per-ray hint.

    python tests/tools/chunk_sim.py [shell|torus] [resolution]

Per ray: every sample the marcher can produce (oracle march_rays), its opacity (oracle geometry chain), the reference's
compositing recurrence -> the composited count; then each schedule is replayed on those opacities.  Test infrastructure
(it imports the oracle); the product never runs it.
"""
import math
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from envidr_amd import scenes                      # noqa: E402
from oracle import clib                            # noqa: E402
from oracle.py import render_oracle as ro          # noqa: E402

TH = 1e-4
F32 = np.float32


def opacities(which: str, R: int):
    scene = scenes.toaster_scene() if which == "shell" else scenes.toaster_scene(shape=scenes.torus(), seed=3)
    rays_o, rays_d = scenes.camera_rays(R, R)
    opt = ro.RenderOptions(ide_mode="exact", visual_items=(), get_normal_image=False)
    o = clib.oracle()
    N = rays_o.shape[0]
    nears, fars = np.empty(N, F32), np.empty(N, F32)
    o.call("near_far_from_aabb", rays_o, rays_d, np.array([-1] * 3 + [1] * 3, F32), N, opt.min_near, nears, fars)
    ns = 256
    xyzs, dirs, deltas = np.zeros((N * ns, 3), F32), np.zeros((N * ns, 3), F32), np.zeros((N * ns, 2), F32)
    o.call("march_rays", N, ns, np.arange(N, dtype=np.int32), nears.copy(), rays_o, rays_d, opt.bound, opt.dt_gamma, opt.max_steps, 1, 128,
           scene.bitfield, nears, fars, xyzs, dirs, deltas, np.zeros(N, F32))
    d = deltas.reshape(N, ns, 2)
    valid = d[:, :, 0] > 0
    idx = np.nonzero(valid.reshape(-1))[0]
    sig = np.zeros(N * ns, F32)
    for i in range(0, len(idx), 200000):
        j = idx[i:i + 200000]
        sig[j] = ro.shade_samples(scene, xyzs[j], dirs[j], opt, 0.0, geometry_only=True)["sigma"]
    return (1 - np.exp(-sig.reshape(N, ns) * d[:, :, 0])).astype(F32), valid


def simulate(alpha, valid, chunk_fn, first=16):
    evaluated = composited = 0
    rounds_hist = []
    for r in range(alpha.shape[0]):
        av = int(valid[r].sum())
        if av == 0:
            continue
        al = alpha[r, :av]
        have = taken = rounds = 0
        ws, done, last = 0.0, False, 0.0
        while not done:
            ch = first if rounds == 0 else chunk_fn(rounds, taken, 1 - ws, last)
            have += max(1, min(ch, av - have))
            rounds += 1
            while taken < have:                       # raymarching.cu:996-1030: stop AFTER the sample whose incoming T is below the threshold
                T = 1 - ws
                ws += al[taken] * T
                last = al[taken]
                taken += 1
                if T < TH:
                    done = True
                    break
            done = done or have >= av
        evaluated += have
        composited += taken
        rounds_hist.append(rounds)
    return evaluated / composited, np.bincount(rounds_hist)


def grow_by_half(rounds, taken, T, a):               # the schedule of rounds 1-2 of this project: 16, then +50 % per round
    return 10 ** 9 if rounds >= 6 else max(8, taken // 2)


def predicted(rounds, taken, T, a, min_chunk=2):     # predicted_chunk() of geometry_pass.hip
    if rounds >= 6:
        return 10 ** 9
    if not (T > TH):
        return 1
    cap = max(8, taken)
    if not (a > 1e-6):
        return cap
    if a >= 1:
        return min_chunk
    j = math.ceil(math.log(TH / T) / math.log(1 - a))
    return int(min(max(j + 2, min_chunk), cap))


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "shell"
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 160
    alpha, valid = opacities(which, R)
    for name, fn in (("grow by half", grow_by_half), ("predicted", predicted)):
        ratio, hist = simulate(alpha, valid, fn)
        print(f"{which} {R}x{R}  {name:13s} evaluated / composited = {ratio:.4f}   rays by rounds needed: {hist.tolist()}")
