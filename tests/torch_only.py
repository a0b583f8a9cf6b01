"""Shared by test_golden_cpu.py (backend "oracle") and test_ops_gpu.py (backend "hip"): the operators against tests/golden/torch_only.npz,
fixtures made from the reference's TORCH-ONLY code (encoding.FreqEncoder, the volume-rendering arithmetic of non_cuda_ray.run; see
tests/golden/make_golden.py golden_torch_only) -- no kernel body and no keyword header takes part in the expected values.

Tolerances, and why they are not zero: the torch code and the CUDA kernels are two formulations of the same quantities in fp32 --
  * FreqEncoder computes sin(x f) and cos(x f); the kernel computes sin(scalbn(x, k) + phase) with phase = pi/2 in fp32 for the cosine
    columns (freqencoder.cu:50-58), i.e. the argument of the cosine columns is rounded once more: |error| <= ulp(|x| 2^k + pi/2) / 2;
  * run() forms the transmittance as cumprod(1 - alpha + 1e-15); the kernels multiply T by (1 - alpha) sample by sample
    (raymarching.cu:567-583, 1009-1030).
"""
from pathlib import Path

import numpy as np

from tests.util import rel_l2, run_op

F = np.float32
GOLD = Path(__file__).parent / "golden" / "torch_only.npz"
FREQ_CASES = [(3, 4), (3, 10), (2, 6), (1, 1)]


def check_freq(backend: str, D: int, deg: int):
    g = np.load(GOLD)
    tag = f"freq_D{D}deg{deg}"
    x, want, gout, want_gx = g[f"{tag}|x"], g[f"{tag}|y"], g[f"{tag}|g"], g[f"{tag}|gx"]
    B, C = x.shape[0], D + 2 * D * deg
    (_, y) = run_op(backend, "freq_encode_forward", x, B, D, deg, C, np.zeros((B, C), F))
    assert np.array_equal(y[:, :D], want[:, :D])                                   # the pass-through columns: exact
    # argument magnitude per column block k: |x| 2^k (+ pi/2 for the cosine columns); one extra fp32 rounding of that argument
    bound = np.zeros(C)
    for k in range(deg):
        arg = 2.0 ** k + np.pi / 2
        bound[D + 2 * D * k: D + 2 * D * (k + 1)] = np.spacing(F(arg)) * 0.5 + 3e-7   # + a few ulp of sin / cos themselves
    err = np.abs(y.astype(np.float64) - want)
    assert (err <= bound[None, :]).all(), (tag, float(err.max()), np.unravel_index(err.argmax(), err.shape))
    assert rel_l2(y, want) <= 2e-5, (tag, rel_l2(y, want))
    # input gradient: sum_k 2^k (g_sin cos - g_cos sin) + g_x -- against torch autograd through the reference module, on the
    # REFERENCE's outputs (the backward kernel reads the forward's outputs, freqencoder.cu:63-94)
    (_, _, gx) = run_op(backend, "freq_encode_backward", gout, want, B, D, deg, C, np.zeros((B, D), F))
    assert rel_l2(gx, want_gx) <= 2e-6, (tag, rel_l2(gx, want_gx))
    return float(err.max()), rel_l2(gx, want_gx)


def _volume_inputs():
    g = np.load(GOLD)
    nears, fars, sigma, rgb = g["vr|nears"], g["vr|fars"], g["vr|sigma"], g["vr|rgb"]
    N, T = sigma.shape
    # the sample positions and steps of non_cuda_ray.py:41-47,119-121 in the same fp32 arithmetic (numpy float32 ops round like torch's)
    lin = np.linspace(0.0, 1.0, T, dtype=F)[None, :]
    z = (nears[:, None] + (fars - nears)[:, None] * lin).astype(F)
    sample_dist = ((fars - nears) / F(T)).astype(F)
    dt = np.concatenate([z[:, 1:] - z[:, :-1], sample_dist[:, None]], axis=1).astype(F)
    deltas = np.stack([dt, z], axis=-1).reshape(N * T, 2).astype(F)               # deltas[:, 1] carries z itself (accum_deltas = 0)
    return g, N, T, z, deltas


def _compare_volume(g, N, ws, depth_wz, image):
    nears, fars, bg = g["vr|nears"], g["vr|fars"], g["vr|bg"]
    want_ws, want_depth, want_image = g["vr|weights_sum"], g["vr|depth"], g["vr|image"]
    assert np.abs(ws - want_ws).max() <= 2e-6, float(np.abs(ws - want_ws).max())
    # the reference's depth is sum w clamp((z - near) / (far - near)); the kernels accumulate sum w z
    depth = (depth_wz.astype(np.float64) - nears * ws.astype(np.float64)) / (fars - nears)
    assert np.abs(depth - want_depth).max() <= 1e-5, float(np.abs(depth - want_depth).max())
    full = image.astype(np.float64) + (1.0 - ws.astype(np.float64))[:, None] * bg[None, :]
    assert rel_l2(full, want_image) <= 1e-6, rel_l2(full, want_image)
    assert np.abs(full - want_image).max() <= 3e-6
    return rel_l2(full, want_image)


def check_composite_train_forward(backend: str):
    g, N, T, z, deltas = _volume_inputs()
    rays = np.stack([np.arange(N), np.arange(N) * T, np.full(N, T)], axis=1).astype(np.int32)
    M = N * T
    out = run_op(backend, "composite_rays_train_forward", g["vr|sigma"].reshape(M), g["vr|rgb"].reshape(M, 3), deltas, rays, M, N, 0.0, 0, 0,
                 np.zeros(N, F), np.zeros(N, F), np.zeros((N, 3), F), np.zeros(M, F))
    ws, depth, image, weights = out[4], out[5], out[6], out[7]
    # per-sample weights against alpha * cumprod(...)[:-1] recomputed from the fixture's inputs in float64 (the fixture stores their sum)
    alpha = 1.0 - np.exp(-deltas[:, 0].reshape(N, T).astype(np.float64) * g["vr|sigma"])
    Tr = np.cumprod(np.concatenate([np.ones((N, 1)), 1.0 - alpha + 1e-15], axis=1), axis=1)[:, :-1]
    assert np.abs(weights.reshape(N, T) - alpha * Tr).max() <= 2e-6
    return _compare_volume(g, N, ws, depth, image)


def check_composite_rays(backend: str):
    """the inference compositor (one call with n_step = T; T_thresh = 0 so that no ray stops early)"""
    g, N, T, z, deltas = _volume_inputs()
    alive = np.arange(N, dtype=np.int32)
    M = N * T
    out = run_op(backend, "composite_rays", N, T, 0.0, 0, 0, alive, g["vr|nears"].copy(), g["vr|sigma"].reshape(M), g["vr|rgb"].reshape(M, 3), deltas,
                 np.zeros(N, F), np.zeros(N, F), np.zeros((N, 3), F))
    ws, depth, image = out[5], out[6], out[7]
    return _compare_volume(g, N, ws, depth, image)
