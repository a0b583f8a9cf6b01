"""TEST TOOLING (build container only: needs oracle/_ref, i.e. /root/reference at build time).

The reference's march_rays kernel body compiled twice -- op by op (oracle/_ref/libenvidr_ref.so) and with a*b+c contracted into
fused multiply-adds the way nvcc compiles device code by default (libenvidr_ref_fma.so) -- on the rays of the benchmark's 800 x 800
frame, marched to the end: how much of the integer trace depends on that choice.

    python tools/contraction_sweep.py [H W] > profiles/r05a/contraction_sweep.txt
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from envidr_amd import scenes  # noqa: E402
from oracle import clib  # noqa: E402
from tests.test_oracle_pinning import _run_on, _voxel_index  # noqa: E402


def main():
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (800, 800)
    plain, fused = clib.ref(), clib.HostLib(clib.REF_LIB.parent / "libenvidr_ref_fma.so", "ref_")
    for name, shape in (("shell (bench scene)", scenes.shell()), ("torus (indirect scene)", scenes.torus())):
        ro, rd = scenes.camera_rays(H, W)
        bitfield = scenes.occupancy_bitfield(shape)
        N = ro.shape[0]
        aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
        z = np.zeros(N, np.float32)
        nf = _run_on(plain, "near_far_from_aabb", (ro, rd, aabb, N, 0.2, z, z))
        nf2 = _run_on(fused, "near_far_from_aabb", (ro, rd, aabb, N, 0.2, z, z))
        nears, fars = nf[3], nf[4]
        nf_diff = int((nf[3] != nf2[3]).sum() + (nf[4] != nf2[4]).sum())
        n_step, chunk = 1024, 8000
        tot = dict(rays=N, samples=0, count_diff_rays=0, time_diff_samples=0, crossed=0, moved=0, max_pos_delta=0.0)
        for lo in range(0, N, chunk):
            alive = np.arange(lo, min(N, lo + chunk), dtype=np.int32)
            n = alive.size
            M = n * n_step + 128
            args = (n, n_step, alive, nears.copy(), ro, rd, 1.0, 0.0, 1024, 1, 128, bitfield, nears, fars,
                    np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32), np.zeros(n, np.float32))
            a, b = _run_on(plain, "march_rays", args), _run_on(fused, "march_rays", args)
            ta, tb = a[-2][: n * n_step], b[-2][: n * n_step]
            ca, cb = (ta[:, 0].reshape(n, n_step) > 0).sum(1), (tb[:, 0].reshape(n, n_step) > 0).sum(1)
            tot["count_diff_rays"] += int((ca != cb).sum())
            both = (ta[:, 0] > 0) & (tb[:, 0] > 0)
            tot["samples"] += int((ta[:, 0] > 0).sum())
            tot["time_diff_samples"] += int((ta[both] != tb[both]).any(axis=1).sum())
            xa, xb = a[-4][: n * n_step][both], b[-4][: n * n_step][both]
            ia = _voxel_index(xa, 1.0, 1, 128, ta[both, 0])
            ib = _voxel_index(xb, 1.0, 1, 128, tb[both, 0])
            tot["crossed"] += int((ia != ib).sum())
            tot["moved"] += int((xa != xb).sum())
            if xa.size:
                tot["max_pos_delta"] = max(tot["max_pos_delta"], float(np.abs(xa.astype(np.float64) - xb).max()))
        print(f"{name}, {H} x {W}: near/far values that differ {nf_diff}; {tot}")


if __name__ == "__main__":
    main()
