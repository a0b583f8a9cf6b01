"""Table gradients of large batches (csrc/hashencoder.hip: k_table_scatter_lds, taken from 2^19 points up): the range-owned LDS
accumulation must produce the sums the per-point atomic kernels produce -- which tests/test_ops_gpu.py checks against the oracle
at the small sizes the oracle finishes in seconds -- and, on a sub-sample of levels, the oracle's own."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(B, seed):
    import torch
    from envidr_amd import scenes
    sc = scenes.toaster_scene()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.rand(B, 3, generator=g)
    x[::1013] = 1.0                      # the cube's far corner (corner + 1 wraps through the modulo)
    x[5::997, 1] = 1.5                   # outside: contributes nothing
    x[7::991] = 0.0
    # half of the points clustered (duplicates inside a wave, the dense levels' hot rows), half uniform
    x[: B // 2] = (x[: B // 2] * 0.05 + 0.4)
    offsets = torch.from_numpy(np.ascontiguousarray(sc.offsets, np.int32))
    S = float(np.log2(sc.per_level_scale))
    return torch, dev, sc, x.to(dev), offsets.to(dev), S, g


@pytest.mark.parametrize("second", [False, True])
def test_lds_scatter_equals_the_per_point_atomics(second):
    from envidr_amd import _lib
    B = (1 << 19) + 777
    torch, dev, sc, x, offsets, S, g = _setup(B, 5)
    L, C, D, H = 16, 2, 3, 16
    rows = int(sc.offsets[L])
    grad = torch.randn(L, B, C, generator=g).to(dev)
    ggx = torch.randn(B, D, generator=g).to(dev)
    dy = torch.zeros(B, L * D * C, device=dev)
    out = torch.empty(L, B, C, device=dev)
    table = torch.from_numpy(sc.table).to(dev)
    _lib.call("hash_encode_forward", x, table, offsets, out, B, D, C, L, S, H, 1, dy)

    def run(lo, hi, into):
        n = hi - lo
        gr = grad[:, lo:hi].contiguous()
        if second:
            gg = torch.zeros(L, n, C, device=dev)
            _lib.call("hash_encode_second_backward", gr, x[lo:hi].contiguous(), table, offsets, n, D, C, L, S, H, 1, dy[lo:hi].contiguous(),
                      ggx[lo:hi].contiguous(), gg, into)
        else:
            _lib.call("hash_encode_backward", gr, x[lo:hi].contiguous(), table, offsets, into, n, D, C, L, S, H, 0, None, None)

    big = torch.zeros(rows, C, device=dev)
    run(0, B, big)                                         # >= 2^19 points: the LDS kernel
    small = torch.zeros(rows, C, device=dev)
    third = B // 3
    for lo, hi in ((0, third), (third, 2 * third), (2 * third, B)):      # < 2^19 points each: per-point atomics, accumulating
        run(lo, hi, small)
    torch.cuda.synchronize()
    a, b = big.cpu().numpy().astype(np.float64), small.cpu().numpy().astype(np.float64)
    assert np.isfinite(a).all()
    # same rows touched, same sums up to the order of fp32 additions (a row collects up to ~10^5 terms on the dense levels).
    # The second gradient adds +v and -v pairs: a row whose terms cancel comes out as exactly 0 in one order of additions and as
    # rounding residue in another, so there the touched-row pattern is only required where the value is not such residue.
    if not second:
        assert np.array_equal(a != 0, b != 0)
    else:
        diff = (a != 0) != (b != 0)
        assert diff.sum() <= 16 and np.abs(np.where(diff, a + b, 0)).max() <= 1e-4 * np.abs(b).max()
    for l in range(L):
        s = slice(int(sc.offsets[l]), int(sc.offsets[l + 1]))
        scale = np.abs(b[s]).max() + 1e-30
        err = np.abs(a[s] - b[s]).max() / scale
        assert err < 2e-4, (l, err)
    assert np.linalg.norm(a - b) / np.linalg.norm(b) < 2e-5


def test_lds_scatter_against_the_oracle_on_two_levels():
    """levels 0 (dense, 4 096 rows) and 1 of a 2-level encoder against the CPU oracle at the size that takes the LDS path"""
    from envidr_amd import _lib
    from tests.util import run_op
    B = (1 << 19) + 131
    torch, dev, sc, x, offsets, S, g = _setup(B, 9)
    L, C, D, H = 2, 2, 3, 16
    offs = offsets[: L + 1].contiguous()
    rows = int(sc.offsets[L])
    grad = torch.randn(L, B, C, generator=g)
    gt = torch.zeros(rows, C, device=dev)
    table = torch.zeros(rows, C, device=dev)
    _lib.call("hash_encode_backward", grad.to(dev), x, table, offs, gt, B, D, C, L, S, H, 0, None, None)
    torch.cuda.synchronize()
    want = run_op("oracle", "hash_encode_backward", grad.numpy(), x.cpu().numpy(), np.zeros((rows, C), np.float32), offs.cpu().numpy(),
                  np.zeros((rows, C), np.float32), B, D, C, L, S, H, 0, None, None)
    ref = want[4]                                          # pointer arguments in order: grad, inputs, embeddings, offsets, grad_embeddings
    assert ref.shape == (rows, C) and np.abs(ref).max() > 0
    got = gt.cpu().numpy()
    assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-4


@pytest.mark.parametrize("D,C", [(3, 1), (3, 4), (3, 8), (2, 1), (2, 2), (2, 4), (2, 8)])
def test_lds_scatter_of_the_other_instantiations(D, C):
    """the (D, C) instantiations of the range-owned kernels besides the toaster encoder's (3, 2), both gradients, against the
    per-point atomic kernels on a small encoder (dense and hashed levels, a level with more rows than one range)"""
    import torch
    from envidr_amd import _lib, scenes
    dev = torch.device("cuda:0")
    L, base, log2T, desired = 5, 8, 16, 1024
    off_np, pls = scenes.hash_level_offsets(D, L, base, log2T, desired)
    offsets = torch.from_numpy(np.ascontiguousarray(off_np, np.int32)).to(dev)
    S = float(np.log2(pls))
    B = (1 << 19) + 321
    g = torch.Generator(device="cpu").manual_seed(100 * D + C)
    x = torch.rand(B, D, generator=g)
    x[::1013] = 1.0
    x[5::997, D - 1] = 1.5
    x[: B // 2] = x[: B // 2] * 0.05 + 0.4
    x = x.to(dev)
    rows = int(off_np[L])
    table = (torch.rand(rows, C, generator=g) * 2 - 1).to(dev)
    grad = torch.randn(L, B, C, generator=g).to(dev)
    ggx = torch.randn(B, D, generator=g).to(dev)
    dy = torch.zeros(B, L * D * C, device=dev)
    _lib.call("hash_encode_forward", x, table, offsets, torch.empty(L, B, C, device=dev), B, D, C, L, S, base, 1, dy)
    for second in ((False, True) if C != 1 else (False,)):
        def run(lo, hi, into):
            n = hi - lo
            gr = grad[:, lo:hi].contiguous()
            if second:
                _lib.call("hash_encode_second_backward", gr, x[lo:hi].contiguous(), table, offsets, n, D, C, L, S, base, 1, dy[lo:hi].contiguous(),
                          ggx[lo:hi].contiguous(), torch.zeros(L, n, C, device=dev), into)
            else:
                _lib.call("hash_encode_backward", gr, x[lo:hi].contiguous(), table, offsets, into, n, D, C, L, S, base, 0, None, None)
        big, small = torch.zeros(rows, C, device=dev), torch.zeros(rows, C, device=dev)
        run(0, B, big)
        third = B // 3
        for lo, hi in ((0, third), (third, 2 * third), (2 * third, B)):
            run(lo, hi, small)
        torch.cuda.synchronize()
        a, b = big.cpu().numpy().astype(np.float64), small.cpu().numpy().astype(np.float64)
        assert np.isfinite(a).all() and np.abs(b).max() > 0
        assert np.linalg.norm(a - b) / np.linalg.norm(b) < 2e-5, (second, np.linalg.norm(a - b) / np.linalg.norm(b))
        for l in range(L):
            s = slice(int(off_np[l]), int(off_np[l + 1]))
            assert np.abs(a[s] - b[s]).max() / (np.abs(b[s]).max() + 1e-30) < 3e-4, (second, l)


def test_mask_scratch_is_sized_to_the_batch_and_can_be_released():
    """the range-mask scratch the LDS scatter keeps per (device, stream) outside torch's allocator: as large as the largest batch seen (a power
    of two from 1 MiB, at most 64 MiB), handed back by envidr_release_scratch() -- and rebuilt by the next call (round-5 advisor finding)"""
    import torch
    from envidr_amd import _lib
    from tests.util import run_op
    _lib.release_scratch()
    rng = np.random.default_rng(3)
    from envidr_amd import scenes
    offsets, pls = scenes.hash_level_offsets()
    L, B = 16, 40000
    table = rng.uniform(-0.1, 0.1, size=(int(offsets[-1]), 2)).astype(np.float32)
    x = rng.uniform(0, 1, size=(B, 3)).astype(np.float32)
    grad = rng.standard_normal((L, B, 2)).astype(np.float32)
    S = float(np.log2(pls))
    args = (grad, x, table, offsets.astype(np.int32), np.zeros_like(table), B, 3, 2, L, S, 16, 0, None, None)
    first = run_op("hip", "hash_encode_backward", *args)[4]
    freed = _lib.release_scratch()
    assert (1 << 20) <= freed <= (4 << 20), freed                  # 16 levels x 40 000 points x 4 B = 2.6 MB of masks -> one 4 MiB buffer, not 64 MiB
    assert _lib.release_scratch() == 0
    again = run_op("hip", "hash_encode_backward", *args)[4]
    assert float(np.abs(first - again).max()) <= 1e-5 * float(np.abs(first).max())
    assert _lib.release_scratch() == freed
