"""Host half of the split-precision mode without a GPU: the (hi, lo) fp16 weight packer of the C-ABI library against a
numpy restatement (np.float16 is IEEE round-to-nearest-even, subnormals included), fragment order and padding."""
import ctypes

import numpy as np
import pytest

from envidr_amd import _lib


def _tile_row(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def _k(order, s, h, i):
    return 16 * s + 8 * h + i if order == 0 else 32 * (s >> 1) + _tile_row(8 * (s & 1) + i, h)


@pytest.fixture(scope="module")
def lib():
    from envidr_amd.fused import _bind_render
    lib = _lib.load()
    _bind_render(lib)
    return lib


@pytest.mark.parametrize("m_out,k_in,order", [(256, 72, 0), (256, 256, 1), (12, 256, 1), (160, 38, 0), (33, 31, 0), (32, 64, 1)])
def test_split_packer_matches_numpy(lib, m_out, k_in, order):
    rng = np.random.default_rng(m_out * 1000 + k_in)
    W = (rng.normal(size=(m_out, k_in)) * rng.choice([1e-6, 1e-3, 0.05, 1.0, 30.0], size=(m_out, k_in))).astype(np.float32)
    W[0, 0], W[-1, -1] = 0.0, -0.0
    n = lib.envidr_split_layer_halves(order, k_in, m_out)
    got = np.zeros(n, np.uint16)
    assert lib.envidr_pack_layer_split(W.ctypes.data, m_out, k_in, order, got.ctypes.data) == 0
    group = lib.envidr_split_group()
    steps = (k_in + 15) // 16 if order == 0 else (k_in + 31) // 32 * 2
    mt = (m_out + 31) // 32
    assert n == steps * mt * 2 * 512
    hi = W.astype(np.float16)
    lo = (W - hi.astype(np.float32)).astype(np.float16)
    want = np.zeros(n, np.uint16)
    frag = 0
    for t0 in range(0, mt, group):
        for s in range(steps):
            for t in range(t0, min(mt, t0 + group)):
                for lane in range(64):
                    for i in range(8):
                        m, k = 32 * t + (lane & 31), _k(order, s, lane >> 5, i)
                        if m < m_out and k < k_in:
                            want[frag * 512 + lane * 8 + i] = hi[m, k].view(np.uint16)
                            want[(frag + 1) * 512 + lane * 8 + i] = lo[m, k].view(np.uint16)
                frag += 2
    assert np.array_equal(got, want)
    # the pair reproduces the weight to ~2^-22 relative (or the fp16 subnormal resolution for tiny values)
    rec = hi.astype(np.float64) + lo.astype(np.float64)
    assert np.all(np.abs(rec - W) <= np.maximum(np.abs(W) * 2.0 ** -21, 2.0 ** -25))


def test_pack_env_split_pads_to_whole_chunks(lib):
    from envidr_amd.fused import pack_env_split
    rng = np.random.default_rng(1)
    env = [(rng.normal(size=(256, 72)), rng.normal(size=256)), (rng.normal(size=(256, 256)), rng.normal(size=256)),
           (rng.normal(size=(256, 256)), rng.normal(size=256)), (rng.normal(size=(12, 256)), rng.normal(size=12))]
    blob, bias = pack_env_split(env)
    assert blob.dtype == np.uint16 and (blob.size * 2) % lib.envidr_split_chunk_bytes() == 0
    assert blob.size * 2 >= (5 * 8 + 16 * 8 + 16 * 8 + 16) * 2 * 1024
    assert bias.dtype == np.float32 and bias.size == (8 + 8 + 8 + 1) * 32


def test_split_packer_rejects_null(lib):
    assert lib.envidr_pack_layer_split(None, 4, 4, 0, None) != 0
    assert b"pack_layer_split" in lib.envidr_last_error()


def test_sdf_geometry_packer_layout(lib):
    """envidr_pack_sdf_geometry (the 16-column geometry kernel's weights) against the layout its header describes"""
    from envidr_amd.fused import pack_sdf_geometry
    rng = np.random.default_rng(3)
    W1, b1 = rng.normal(size=(64, 32)).astype(np.float32), rng.normal(size=64).astype(np.float32)
    W2, b2 = rng.normal(size=(64, 64)).astype(np.float32), rng.normal(size=64).astype(np.float32)
    W3, b3 = rng.normal(size=(15, 64)).astype(np.float32), rng.normal(size=15).astype(np.float32)
    blob = pack_sdf_geometry([(W1, b1), (W2, b2), (W3, b3)])
    assert blob.size == lib.envidr_sdf_geometry_floats() and blob.size % 64 == 0
    frag = blob[:-64].reshape(-1, 64)
    assert np.array_equal(blob[-64:], W3[0])
    L1, L2, L3, B2, B1 = 0, 36, 36 + 68, 36 + 68 + 17, 36 + 68 + 17 + 64
    w3_row = lambda q, r: ({0: 0, 1: 13, 2: 14, 3: -1}[q] if r == 0 else 1 + 3 * q + (r - 1))
    for lane in (0, 5, 17, 38, 63):
        m, kq = lane & 15, lane >> 4
        for T in range(4):
            assert frag[L1 + T, lane] == (b1[16 * T + m] if kq == 0 else 0)
            for s in range(8):
                assert frag[L1 + (1 + s) * 4 + T, lane] == W1[16 * T + m, 2 * (4 * (s >> 1) + kq) + (s & 1)]
            for s in range(16):
                k = 16 * (s >> 2) + 4 * kq + (s & 3)
                assert frag[L2 + (1 + s) * 4 + T, lane] == W2[16 * T + m, k]
                assert frag[B2 + s * 4 + T, lane] == W2[k, 16 * T + m]
        row = w3_row(m >> 2, m & 3)
        assert frag[L3, lane] == (b3[row] if kq == 0 and row >= 0 else 0)
        for s in range(16):
            k = 16 * (s >> 2) + 4 * kq + (s & 3)
            assert frag[L3 + 1 + s, lane] == (W3[row, k] if row >= 0 else 0)
            for T in range(2):
                q, r = m >> 2, m & 3
                assert frag[B1 + s * 2 + T, lane] == W1[k, 2 * (4 * (2 * T + (r >> 1)) + q) + (r & 1)]
    assert np.all(frag[217:] == 0)
