"""The BENCHMARKED hash path (csrc/hash_lean.hip.h inside k_geo_eval32) pinned as integers and per value.

The standalone operator (`hash_encode_forward`, csrc/grid_core.hip.h) is bit-exact against the oracle; the geometry kernel
that every frame of the pipeline -- and bench.py's headline -- runs gathers with a leaner restatement of the same function.
These tests compare THAT kernel, through `envidr_geometry_probe`, with the reference's index arithmetic
(hashencoder/src/hashencoder.cu:36-69 fast_hash / get_grid_index, :124-149 out-of-range rule, :160-190 cell + corners):

  * the table row of every (sample, level, corner) EQUALS the oracle's -- dense levels (x + y res + z res^2, `% size` on
    sizes that are not powers of two), hashed levels (prime XOR, `& (2^19 - 1)`), the wrap of the +1 corners at x = 1.0,
    cube faces / edges / corners, cell boundaries of every level;
  * the 32 features agree with `hash_encode_forward` to fp32 rounding of the corner values (the lean path factorises
    the trilinear sum: measured 3.1, bound 4 ulp of the largest corner magnitude);
  * raw SDF-network outputs, geo_feat VALUES (not just their norm), sdf, blend and the unnormalised gradient agree with
    the oracle's chain per sample; samples whose hidden pre-activations sit within fp32 rounding of a ReLU kink are
    COUNTED (they are the only ones allowed a different gradient) instead of being covered by a looser bound.
"""
import ctypes

import numpy as np
import pytest

from envidr_amd import scenes
from tests.util import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    return scenes.toaster_scene()


@pytest.fixture(scope="module")
def renderer(scene):
    from envidr_amd.fused import FusedRenderer
    return FusedRenderer.from_scene(scene)


def level_scales(scene, H=16):
    """exp2f(level * S) * H - 1.0f (hashencoder.cu:152) with libm's exp2f, like the oracle and the library's host side"""
    libm = ctypes.CDLL("libm.so.6")
    libm.exp2f.argtypes, libm.exp2f.restype = [ctypes.c_float], ctypes.c_float
    S = np.float32(np.log2(scene.per_level_scale))
    return [np.float32(np.float32(libm.exp2f(float(np.float32(l) * S))) * np.float32(H) - np.float32(1.0)) for l in range(16)]


def probe_points(scene, n_random: int, seed: int = 11) -> np.ndarray:
    """positions in [-1, 1]^3 (bound 1): random, the shell the benchmark samples, every face / edge / corner pattern of the cube,
    x01 = 1.0 exactly, cell boundaries of every level (x01 = k / scale_l and its fp32 neighbours), points outside"""
    rng = np.random.default_rng(seed)
    parts = [rng.uniform(-1, 1, size=(n_random, 3))]
    d = rng.normal(size=(n_random // 2, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    parts.append(d * rng.uniform(0.44, 0.56, size=(n_random // 2, 1)))
    # faces, edges, corners: each coordinate independently -1, +1 or free
    pat = np.array(np.meshgrid([-1, 0, 1], [-1, 0, 1], [-1, 0, 1], indexing="ij")).reshape(3, -1).T
    for p in pat:
        if not p.any():
            continue
        q = rng.uniform(-1, 1, size=(512, 3))
        q[:, p != 0] = p[p != 0]
        parts.append(q)
    # cell boundaries: x01 = k / scale for a few k per level, +- one fp32 step, on every axis in turn
    for l, sc in enumerate(level_scales(scene)):
        res = int(np.ceil(sc)) + 1
        ks = np.unique(np.concatenate([rng.integers(0, res, size=24), [0, 1, res - 2, res - 1]])).astype(np.float64)
        x01 = (ks / np.float64(sc)).astype(np.float32)
        x01 = np.concatenate([x01, np.nextafter(x01, np.float32(2)), np.nextafter(x01, np.float32(-1))])
        x01 = x01[(x01 >= 0) & (x01 <= 1)]
        for axis in range(3):
            q = rng.uniform(-1, 1, size=(x01.size, 3))
            q[:, axis] = x01.astype(np.float64) * 2 - 1
            parts.append(q)
    out = rng.uniform(-1.6, 1.6, size=(2048, 3))
    parts.append(out[np.any(np.abs(out) > 1, axis=1)])
    xyz = np.concatenate(parts).astype(np.float32)
    xyz[:4] = [[1, 1, 1], [-1, -1, -1], [1, -1, 1], [0, 0, 0]]
    return np.ascontiguousarray(xyz)


def oracle_rows(xyz01: np.ndarray, scene) -> np.ndarray:
    from oracle import clib
    lib = clib.oracle().lib
    B = xyz01.shape[0]
    rows = np.empty((B, 16, 8), np.uint32)
    offs = np.ascontiguousarray(scene.offsets, np.int32)
    fn = lib.oracle_hash_corner_rows
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_float, ctypes.c_uint32]
    fn.restype = ctypes.c_int
    assert fn(xyz01.ctypes.data, offs.ctypes.data, rows.ctypes.data, B, 16, float(np.log2(scene.per_level_scale)), 16) == 0
    return rows


def test_corner_rows_and_features_of_the_frame_kernel_equal_the_oracle(scene, renderer):
    import torch
    from oracle import clib
    xyz = probe_points(scene, 700_000)
    M = xyz.shape[0]
    assert M >= 1_000_000
    x01 = ((xyz + np.float32(1.0)) / np.float32(2.0)).astype(np.float32)        # hashgrid.py:161 with bound = 1
    inside = np.all((x01 >= 0) & (x01 <= 1), axis=1)
    assert inside.sum() > 0.99 * M - 4096 and (~inside).sum() > 500
    got = renderer.geometry_probe(torch.from_numpy(xyz).cuda(), want=("features", "corner_rows"))
    torch.cuda.synchronize()
    rows = got["corner_rows"].cpu().numpy().view(np.uint32)
    want_rows = oracle_rows(x01, scene)
    # --- integers: every row of every inside point, all 16 levels x 8 corners
    bad = rows[inside] != want_rows[inside]
    assert not bad.any(), (int(bad.sum()), np.argwhere(bad)[:5].tolist())
    # the dense / hashed decision per level, independently in numpy (hashencoder.cu:55-70: linear while the strides fit the
    # level, else the prime hash; `% size` in both cases): levels 0-4 of this table are dense with non-power-of-two sizes
    sizes = np.diff(scene.offsets).astype(np.int64)
    xin = x01[inside][:4096]
    n_dense = 0
    for l, sc in enumerate(level_scales(scene)):
        res = int(np.ceil(sc)) + 1
        c = np.floor(xin * sc).astype(np.int64)
        for corner in (0, 7):
            q = c + np.array([(corner >> d) & 1 for d in range(3)])
            if res ** 3 <= sizes[l]:
                expect = (q[:, 0] + q[:, 1] * res + q[:, 2] * res * res) % sizes[l]
                n_dense += corner == 0
            else:
                h = (q[:, 0].astype(np.uint32) * np.uint32(1)) ^ (q[:, 1].astype(np.uint32) * np.uint32(2654435761)) ^ (q[:, 2].astype(np.uint32) * np.uint32(805459861))
                expect = h.astype(np.int64) % sizes[l]
            assert np.array_equal(expect, rows[inside][:4096, l, corner].astype(np.int64)), (l, corner)
    assert n_dense == 5 and any(s_ & (s_ - 1) for s_ in sizes[:5])
    # --- values: features vs the bit-exact standalone operator's definition (the oracle's hash_encode_forward)
    out = np.empty((16, M, 2), np.float32)
    clib.oracle().call("hash_encode_forward", x01, scene.table, np.ascontiguousarray(scene.offsets, np.int32), out, M, 3, 2, 16,
                       float(np.log2(scene.per_level_scale)), 16, 0, None)
    want_feat = out.transpose(1, 0, 2).reshape(M, 32)
    feat = got["features"].cpu().numpy()
    assert np.all(feat[~inside] == 0) and np.all(want_feat[~inside] == 0)        # hashencoder.cu:124-149
    ulp = np.float32(np.abs(scene.table).max()) * np.float32(2.0 ** -23)
    err = np.abs(feat.astype(np.float64) - want_feat)
    assert err.max() <= 4.0 * ulp, (err.max() / ulp)          # measured 3.1 (1 M points): the factorised sum rounds at other places than the expanded one
    assert rel_l2(feat, want_feat) <= 2e-7


def test_network_outputs_of_the_frame_kernel_match_the_oracle_chain_per_sample(scene, renderer):
    """sdf, geo_feat (values), blend, roughness and the sdf gradient per sample; ReLU-kink samples counted, not averaged away"""
    import torch
    from oracle.py import render_oracle as ro
    xyz = probe_points(scene, 60_000, seed=12)
    M = xyz.shape[0]
    inside = np.all(np.abs(xyz) <= 1, axis=1)
    dirs = np.tile(np.array([[0, 0, 1]], np.float32), (M, 1))
    want = ro.shade_samples(scene, xyz, dirs, ro.RenderOptions(ide_mode="exact"), None, geometry_only=True)
    got = renderer.geometry_probe(torch.from_numpy(xyz).cuda())
    ev = renderer.geometry_eval(torch.from_numpy(xyz).cuda(), want=("sigma", "normal", "geo_feat", "roughness", "blend"))
    torch.cuda.synchronize()
    raw = got["raw_outputs"].cpu().numpy()
    sdf = raw[:, 0]
    assert np.abs(sdf - want["sdf"]).max() <= 1e-6 + 1e-5 * np.abs(want["sdf"]).max()
    geo = ev["geo_feat"].cpu().numpy()
    assert np.abs(geo[inside] - want["geo_feat"][inside]).max() <= 1e-5                       # values, every component
    g_raw = raw[:, 1:13] / np.maximum(np.linalg.norm(raw[:, 1:13], axis=1, keepdims=True), 1e-12)
    assert np.abs(g_raw[inside] - want["geo_feat"][inside]).max() <= 1e-5
    assert np.abs(1 / (1 + np.exp(-raw[:, 14].astype(np.float64))) - want["blend"].reshape(-1)).max() <= 1e-5
    assert np.array_equal(ev["blend"].cpu().numpy(), raw[:, 14])
    assert rel_l2(ev["roughness"].cpu().numpy(), want["roughness"].reshape(-1)) <= 1e-5
    assert rel_l2(ev["sigma"].cpu().numpy(), want["sigma"]) <= 1e-5
    # --- the gradient.  Hidden pre-activations in fp64 from the GPU's own features: a sample is "on a kink" when one of its 128
    # hidden units is within fp32 rounding of zero -- only there may two correct fp32 evaluations pick different ReLU masks
    feat = got["features"].cpu().numpy().astype(np.float64)
    (W1, b1), (W2, b2), (W3, b3) = [(W.astype(np.float64), b.astype(np.float64)) for W, b in scene.mlps["sdf"]]
    h1 = feat @ W1.T + b1
    h2 = np.maximum(h1, 0) @ W2.T + b2
    margin = np.minimum(np.abs(h1).min(axis=1) / np.maximum(np.abs(h1).max(axis=1), 1e-30),
                        np.abs(h2).min(axis=1) / np.maximum(np.abs(h2).max(axis=1), 1e-30))
    kink = margin < 4e-6
    grad = got["sdf_gradient"].cpu().numpy().astype(np.float64)
    # the oracle's autograd gradient, unnormalised: recompute from its normal's definition through the same chain
    import torch as T
    x = T.from_numpy(xyz).requires_grad_(True)
    f = ro.hash_encode(x, scene, ro.RenderOptions())
    s = ro._mlp(scene.mlps["sdf"], f)[..., 0]
    gref = T.autograd.grad(s, x, T.ones_like(s))[0].numpy().astype(np.float64)
    sel = inside & ~kink
    gerr = np.linalg.norm(grad[sel] - gref[sel], axis=1) / np.maximum(np.linalg.norm(gref[sel], axis=1), 1e-6)
    assert gerr.max() <= 2e-4, (gerr.max(), int(np.argmax(gerr)))
    assert np.quantile(gerr, 0.99) <= 2e-5
    # how many samples COULD flip (128 hidden units x a relative margin of 4e-6: ~1e-3 of the samples) ...
    n_kink = int((kink & inside).sum())
    assert n_kink <= int(2e-3 * M), n_kink
    # ... and how many DID: a flipped ReLU mask changes the gradient in its leading digits.  This count -- not a looser bound on
    # everybody -- is what the cross-implementation frame tests allow for (tests/test_geometry_gpu.py, test_dropin_gpu.py)
    k_err = np.linalg.norm(grad[kink & inside] - gref[kink & inside], axis=1) / np.maximum(np.linalg.norm(gref[kink & inside], axis=1), 1e-6)
    n_flipped = int((k_err > 1e-3).sum())
    print(f"ReLU kinks: {n_kink} of {M} samples within the margin, {n_flipped} with a flipped mask (gradient off by > 1e-3)")
    assert n_flipped <= max(4, int(1e-4 * M)), (n_flipped, n_kink)
    # and away from kinks the normals agree per sample, not just on average
    n_err = np.abs(ev["normal"].cpu().numpy()[sel] - want["normal"][sel]).max(axis=1)
    gn = np.linalg.norm(gref[sel], axis=1)
    assert (n_err[gn > 1e-2] <= 2e-4).all(), float(n_err[gn > 1e-2].max())


def test_probe_argument_checks(renderer):
    import torch
    from envidr_amd import _lib
    x = torch.zeros(4, 3, device="cuda")
    assert renderer.geometry_probe(x[:0])["features"].shape == (0, 32)
    with pytest.raises(_lib.EnvidrError):
        renderer.geometry_probe(x, want=())
