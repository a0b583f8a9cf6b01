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


@pytest.mark.parametrize("ide_degree,hidden", [(5, 256), (4, 160)])
def test_two_group_packer_matches_its_layout(lib, ide_degree, hidden):
    """envidr_pack_env_split2 (csrc/mlp_split2.hip.h Split2Layout): per layer-1 tile t [its S1 steps | steps 2t, 2t+1 of every layer-2 tile],
    then per layer-3 tile t [its 2 T steps | steps 2t, 2t+1 of the 12-row last layer] -- in the kernel's consumption order, in which the
    tile after next is computed before a tile is consumed (A1(0) A1(1) | A2(0) A1(2) | ...; B3(0) | B3(1) B4(0) | ...); layer 1 in lane order,
    the rest in tile order; (hi, lo) fragments of 512 halves each; padded to whole 8-fragment chunks"""
    from envidr_amd.fused import pack_env_split2
    terms = 2 ** ide_degree - 1 + ide_degree
    K1, T = 2 * terms, hidden // 32
    S1, SH = (K1 + 15) // 16, 2 * T
    rng = np.random.default_rng(ide_degree)
    env = [(rng.normal(size=(hidden, K1)), rng.normal(size=hidden)), (rng.normal(size=(hidden, hidden)), rng.normal(size=hidden)),
           (rng.normal(size=(hidden, hidden)), rng.normal(size=hidden)), (rng.normal(size=(12, hidden)), rng.normal(size=12))]
    env = [(W.astype(np.float32) * rng.choice([1e-5, 0.05, 1.0, 20.0], size=W.shape).astype(np.float32), b.astype(np.float32)) for W, b in env]
    blob, bias = pack_env_split2(env, ide_degree)
    frags = T * (2 * S1 + 4 * T) + T * (2 * SH + 4)
    padded = (frags + 7) // 8 * 8
    assert blob.dtype == np.uint16 and blob.size == padded * 512 == lib.envidr_env_split2_halves(ide_degree, hidden)
    assert bias.dtype == np.float32 and bias.size == (3 * T + 1) * 32
    F = blob.reshape(padded, 64, 8)
    (W1, _), (W2, _), (W3, _), (W4, _) = env
    hi = lambda W: W.astype(np.float16)
    lo = lambda W: (W - W.astype(np.float16).astype(np.float32)).astype(np.float16)
    A1, A2, B3, B4 = 2 * S1, 4 * T, 2 * SH, 4
    FB = T * (A1 + A2)
    a2_block = lambda t: min(t + 2, T) * A1 + t * A2
    a1_block = lambda t: t * A1 if t < 2 else a2_block(t - 2) + A2
    b4_block = lambda t: FB + min(t + 2, T) * B3 + t * B4
    b3_block = lambda t: FB if t == 0 else FB + t * B3 + (t - 1) * B4
    # the blocks tile the pass without gaps or overlaps, in execution order
    order = [("a1", 0), ("a1", 1)] + [x for t in range(T) for x in ([("a2", t)] + ([("a1", t + 2)] if t + 2 < T else []))]
    order += [("b3", 0)] + [x for t in range(T) for x in (([("b3", t + 1)] if t + 1 < T else []) + [("b4", t)])]
    at = 0
    for kind, t in order:
        start, size = {"a1": (a1_block, A1), "a2": (a2_block, A2), "b3": (b3_block, B3), "b4": (b4_block, B4)}[kind]
        assert start(t) == at, (kind, t)
        at += size
    assert at == frags
    rows = np.arange(32)

    def check(frag, W, m_of_lane, k_of):
        for h in (0, 1):
            for i in range(8):
                k = k_of(h, i)
                m = m_of_lane
                ok = (m < W.shape[0]) & (k >= 0)
                want_h = np.where(ok, hi(W)[np.minimum(m, W.shape[0] - 1), max(k, 0)], np.float16(0)).view(np.uint16)
                want_l = np.where(ok, lo(W)[np.minimum(m, W.shape[0] - 1), max(k, 0)], np.float16(0)).view(np.uint16)
                assert np.array_equal(F[frag, 32 * h:32 * h + 32, i], want_h), (frag, h, i)
                assert np.array_equal(F[frag + 1, 32 * h:32 * h + 32, i], want_l), (frag, h, i)

    for t in (0, T // 2, T - 1):
        for s in range(S1):
            def k1(h, i):
                k = _k(0, s, h, i)
                return -1 if k >= K1 else k
            check(a1_block(t) + 2 * s, W1, 32 * t + rows, k1)
        for s in (0, 1):
            for u in (0, T - 1):
                check(a2_block(t) + s * 2 * T + 2 * u, W2, 32 * u + rows, lambda h, i: _k(1, 2 * t + s, h, i))
        for s in (0, 1, SH - 1):
            check(b3_block(t) + 2 * s, W3, 32 * t + rows, lambda h, i: _k(1, s, h, i))
        for s in (0, 1):
            check(b4_block(t) + 2 * s, W4, rows, lambda h, i: _k(1, 2 * t + s, h, i))
    assert not F[frags:].any()
    with pytest.raises(_lib.EnvidrError):
        pack_env_split2([(np.zeros((128, 72), np.float32), np.zeros(128, np.float32))] * 4, 5)


@pytest.mark.parametrize("ide_degree,hidden", [(5, 256), (4, 160)])
def test_fused_pair_blob_reproduces_the_mlp_under_the_kernels_dataflow(lib, ide_degree, hidden):
    """A numpy model of k_env_split2's dataflow eats the packed blob fragment by fragment, in stream order, with the matrix instruction's
    contract -- D[m][n] += sum over (half h, slot i) of A[lane (m, h)][i] * B[lane (n, h)][i] -- and the kernel's operand conventions (layer 1:
    slot (s, h, i) = input 16 s + 8 h + i; later layers: the producing tile's accumulator register 8 (s & 1) + i of lane half h, i.e. row
    tile_row(8 (s & 1) + i, h) of tile s >> 1): phase A = layer-1 tile t, then its two steps of every layer-2 tile; in-place (hi, lo) split;
    phase B = layer-3 tile t, then its two steps of the last layer.  The result must be the MLP evaluated with (hi, lo) operands."""
    from envidr_amd.fused import pack_env_split2
    terms = 2 ** ide_degree - 1 + ide_degree
    K1, T = 2 * terms, hidden // 32
    S1, SH = (K1 + 15) // 16, 2 * T
    rng = np.random.default_rng(7 + ide_degree)
    dims = [K1, hidden, hidden, hidden, 12]
    env = [((rng.normal(size=(dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32), (0.1 * rng.normal(size=dims[i + 1])).astype(np.float32))
           for i in range(4)]
    blob, _ = pack_env_split2(env, ide_degree)
    F = blob.view(np.float16).astype(np.float64).reshape(-1, 64, 8)                     # [fragment][lane][slot]
    A1, A2, B3, B4 = 2 * S1, 4 * T, 2 * SH, 4
    FB = T * (A1 + A2)
    a2_block = lambda t: min(t + 2, T) * A1 + t * A2
    a1_block = lambda t: t * A1 if t < 2 else a2_block(t - 2) + A2
    b4_block = lambda t: FB + min(t + 2, T) * B3 + t * B4
    b3_block = lambda t: FB if t == 0 else FB + t * B3 + (t - 1) * B4
    split = lambda v: ((hi := v.astype(np.float16).astype(np.float64)), (v - hi).astype(np.float16).astype(np.float64))
    lanes_m, lanes_h = np.arange(64) & 31, np.arange(64) >> 5

    def mfma(acc, frag, Bh, Bl):
        """acc [32 rows][32 items] += the three products of one (hi, lo) fragment pair with the B operand pair ([64 lanes][8 slots] each)"""
        Ah, Al = F[frag], F[frag + 1]
        for A, B in ((Ah, Bh), (Ah, Bl), (Al, Bh)):
            for h in (0, 1):
                acc += A[lanes_h == h] @ B[lanes_h == h].T                               # [32 m][8] x [8][32 n]

    def operand_from_rows(rows_hi, rows_lo, s):
        """B operand pair of step s from a finished tile's rows [32 rows][32 items]: slot (h, i) = row tile_row(8 (s & 1) + i, h)"""
        Bh, Bl = np.zeros((64, 8)), np.zeros((64, 8))
        for h in (0, 1):
            for i in range(8):
                row = _tile_row(8 * (s & 1) + i, h)
                Bh[32 * h:32 * h + 32, i], Bl[32 * h:32 * h + 32, i] = rows_hi[row], rows_lo[row]
        return Bh, Bl

    x = rng.normal(size=(32, K1)).astype(np.float32)                                     # 32 items
    xh, xl = split(x.astype(np.float64))
    bias = [b.astype(np.float64) for _, b in env]
    relu_split = lambda acc: split(np.clip(acc, 0.0, 60000.0))
    # phase A
    acc2 = [np.tile(bias[1][32 * u:32 * u + 32, None], (1, 32)) for u in range(T)]
    for t in range(T):
        acc1 = np.tile(bias[0][32 * t:32 * t + 32, None], (1, 32))
        for s in range(S1):
            Bh, Bl = np.zeros((64, 8)), np.zeros((64, 8))
            for h in (0, 1):
                for i in range(8):
                    k = 16 * s + 8 * h + i
                    if k < K1:
                        Bh[32 * h:32 * h + 32, i], Bl[32 * h:32 * h + 32, i] = xh[:, k], xl[:, k]
            mfma(acc1, a1_block(t) + 2 * s, Bh, Bl)
        yh, yl = relu_split(acc1)
        for s in (0, 1):
            Bh, Bl = operand_from_rows(yh, yl, s)
            for u in range(T):
                mfma(acc2[u], a2_block(t) + s * 2 * T + 2 * u, Bh, Bl)
    x2 = [relu_split(a) for a in acc2]
    # phase B
    acc4 = np.tile(np.concatenate([bias[3], np.zeros(20)])[:, None], (1, 32))
    for t in range(T):
        acc3 = np.tile(bias[2][32 * t:32 * t + 32, None], (1, 32))
        for s in range(SH):
            Bh, Bl = operand_from_rows(*x2[s >> 1], s)
            mfma(acc3, b3_block(t) + 2 * s, Bh, Bl)
        zh, zl = relu_split(acc3)
        for s in (0, 1):
            Bh, Bl = operand_from_rows(zh, zl, s)
            mfma(acc4, b4_block(t) + 2 * s, Bh, Bl)
    got = acc4[:12].T                                                                     # [item][12]
    h = x.astype(np.float64)
    for i, (W, b) in enumerate(env):
        h = h @ W.astype(np.float64).T + b.astype(np.float64)
        if i < 3:
            h = np.maximum(h, 0.0)
    assert np.abs(got - h).max() <= 2e-6 * max(1.0, float(np.abs(h).max())), float(np.abs(got - h).max())
    assert not acc4[12:].any()                                                            # the padded rows of the 12-row layer stay zero
