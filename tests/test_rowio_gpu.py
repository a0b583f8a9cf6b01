"""The streaming operators' coalesced paths (csrc/rowio.hip.h: row tiles through LDS, 16-byte accesses; k_composite_rays_vec) against
their plain forms: ragged batch sizes, and arrays that are NOT 16-byte aligned (views one float into a buffer), which take the
4-byte fallbacks -- both must give the bits of the aligned call, which tests/test_ops_gpu.py checks against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _shifted(t):
    """the same values in storage that starts 4 bytes off a 16-byte boundary"""
    import torch
    buf = torch.empty(t.numel() + 8, dtype=t.dtype, device=t.device)
    v = buf[1:1 + t.numel()].view(t.shape)
    v.copy_(t)
    assert v.data_ptr() % 16 == 4
    return v


@pytest.mark.parametrize("B", [1, 63, 64, 65, 1000, 4097])
def test_encoders_ragged_and_unaligned(B):
    import torch
    from envidr_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B)
    d = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1).to(dev)
    rough = torch.rand(B, generator=g).to(dev)
    # spherical harmonics (+ dy_dx), frequency (forward / backward), IDE
    for deg in range(1, 9):          # every degree: each is its own instantiation, aligned and not
        C2 = deg * deg
        o, dy = torch.zeros(B, C2, device=dev), torch.zeros(B, 3 * C2, device=dev)
        _lib.call("sh_encode_forward", d, o, B, 3, deg, dy)
        o2, dy2 = _shifted(torch.zeros(B, C2, device=dev)), _shifted(torch.zeros(B, 3 * C2, device=dev))
        _lib.call("sh_encode_forward", d, o2, B, 3, deg, dy2)
        assert torch.equal(o, o2) and torch.equal(dy, dy2), ("sh", deg)
        o3 = _shifted(torch.zeros(B, C2, device=dev))
        _lib.call("sh_encode_forward", d, o3, B, 3, deg, None)          # unaligned outputs without the Jacobian: its own instantiation
        assert torch.equal(o, o3), ("sh without dy_dx", deg)
        # reference values: one point at a time through the same operator (a single-row tile)
        one = torch.zeros(1, C2, device=dev)
        _lib.call("sh_encode_forward", d[B - 1:B].contiguous(), one, 1, 3, deg, None)
        assert torch.equal(one[0], o[B - 1])
        gr = torch.randn(B, C2, generator=g).to(dev)
        gi, gi2 = torch.zeros(B, 3, device=dev), torch.zeros(B, 3, device=dev)
        _lib.call("sh_encode_backward", gr, d, B, 3, deg, dy, gi)
        _lib.call("sh_encode_backward", _shifted(gr), d, B, 3, deg, _shifted(dy), gi2)
        assert torch.equal(gi, gi2), ("sh backward", deg)
    for deg in range(1, 11):
        C = 3 + 6 * deg
        o = torch.zeros(B, C, device=dev)
        _lib.call("freq_encode_forward", d, B, 3, deg, C, o)
        o2 = _shifted(torch.zeros(B, C, device=dev))
        _lib.call("freq_encode_forward", d, B, 3, deg, C, o2)
        assert torch.equal(o, o2), ("freq", deg)
        gr = torch.randn(B, C, generator=g).to(dev)
        gi, gi2 = torch.zeros(B, 3, device=dev), _shifted(torch.zeros(B, 3, device=dev))
        _lib.call("freq_encode_backward", gr, o, B, 3, deg, C, gi)
        _lib.call("freq_encode_backward", _shifted(gr), _shifted(o), B, 3, deg, C, gi2)
        assert torch.equal(gi, gi2), ("freq backward", deg)
    for deg in (1, 2, 3, 4, 5):
        C = 2 * ((1 << deg) - 1 + deg)
        o, o2 = torch.zeros(B, C, device=dev), _shifted(torch.zeros(B, C, device=dev))
        _lib.call("ide_encode_forward", d, rough, 0.0, B, deg, o)
        _lib.call("ide_encode_forward", d, rough, 0.0, B, deg, o2)
        assert torch.equal(o, o2) and torch.isfinite(o).all(), ("ide", deg)


@pytest.mark.parametrize("n_step", [4, 8])
def test_composite_rays_vector_loads_equal_the_scalar_kernel(n_step):
    import torch
    from envidr_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n_step)
    N = 5003
    M = N * n_step
    sig = (torch.rand(M, generator=g) * 60).to(dev)
    rgb = torch.rand(M, 3, generator=g).to(dev)
    dl = torch.full((M, 2), 0.0034).to(dev)
    dl[::37, 0] = 0                                        # exhausted samples in the middle of some rays
    order = torch.randperm(N, generator=g).to(torch.int32)

    def run(shift):
        alive = order.clone().to(dev)
        rt = torch.zeros(N, device=dev)
        ws, dp, im = torch.zeros(N, device=dev), torch.zeros(N, device=dev), torch.zeros(N, 3, device=dev)
        s, c, d = (sig, rgb, dl) if not shift else (_shifted(sig), _shifted(rgb), _shifted(dl))
        _lib.call("composite_rays", N, n_step, 1e-4, 1, 0, alive, rt, s, c, d, ws, dp, im)
        torch.cuda.synchronize()
        return alive, rt, ws, dp, im
    a, b = run(False), run(True)                           # aligned: 16-byte loads; shifted: the scalar kernel
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert int((a[0] < 0).sum()) > 0 and int((a[0] >= 0).sum()) > 0
