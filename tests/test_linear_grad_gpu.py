"""envidr_linear_weight_grad (csrc/linear_grad.hip): the weight / bias gradient of a dense layer over a large batch, against float64 sums;
and the autograd Function that uses it in the training branch against nn.Linear's own gradients."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K,N", [(1, 3, 5), (2, 64, 64), (127, 72, 256), (4097, 256, 256), (20000, 256, 12), (33333, 24, 32), (50001, 28, 64),
                                   (16384, 64, 3), (999, 4, 64), (70000, 160, 160), (65, 38, 160), (3000, 100, 132)])
def test_weight_gradient_matches_float64_sums(M, K, N):
    """every operand form (wide = 16-byte loads / 4 tiles per block, narrow), ragged widths, odd sample counts, chunks of any size"""
    import torch
    from envidr_amd.fused import linear_weight_grad
    g = torch.Generator(device="cuda").manual_seed(M + 7 * K + 13 * N)
    x = torch.randn(M, K, device="cuda", generator=g)
    gy = torch.randn(M, N, device="cuda", generator=g)
    dW, db = linear_weight_grad(x, gy)
    want_W = (gy.double().t() @ x.double())
    want_b = gy.double().sum(0)
    scale = float(np.sqrt(M))                          # magnitude of a sum of M products of unit normals
    assert float((dW.double() - want_W).abs().max()) <= 2e-6 * scale * max(1.0, np.log2(M + 1))
    assert float((db.double() - want_b).abs().max()) <= 2e-6 * scale * max(1.0, np.log2(M + 1))
    dW2, none = linear_weight_grad(x, gy, bias=False)
    assert none is None and torch.equal(dW, dW2)       # deterministic: partial sums are added in a fixed order


def test_big_batch_linear_has_nn_linear_gradients_to_second_order():
    """`_linear` in the training branch (x W^T + b with the split weight gradient) against nn.Linear: first-order gradients, and the
    eikonal pattern -- a gradient w.r.t. the input taken with create_graph, squared, and differentiated again w.r.t. weights and input"""
    import torch
    from envidr_amd.nerf import network as nw
    torch.manual_seed(0)
    l1, l2 = torch.nn.Linear(32, 64).cuda().train(), torch.nn.Linear(64, 15).cuda().train()
    x0 = torch.randn(20000, 32, device="cuda")
    w = torch.randn(20000, 15, device="cuda")

    def run(linear):
        for l in (l1, l2):
            l.zero_grad()
        x = x0.clone().requires_grad_(True)
        h = torch.relu(linear(l1, x))
        y = linear(l2, h)
        gx, = torch.autograd.grad(y[:, 0].sum(), x, create_graph=True)            # "normals"
        loss = (y * w).sum() + ((gx.norm(dim=-1) - 1) ** 2).sum() + (gx * x0[:, :32]).sum()
        loss.backward()
        return [x.grad.clone(), l1.weight.grad.clone(), l1.bias.grad.clone(), l2.weight.grad.clone(), l2.bias.grad.clone(), gx.detach().clone()]
    got = run(nw._linear)
    want = run(lambda l, h: l(h))
    for a, b in zip(got, want):
        assert float((a - b).norm() / b.norm()) <= 5e-6, float((a - b).norm() / b.norm())
    # the weight gradients really went through the operator (4 first-order + second-order calls), and below the row threshold or with
    # grad disabled the layer IS nn.Linear
    calls = []
    real = nw._fused.linear_weight_grad
    nw._fused.linear_weight_grad = lambda *a, **k: (calls.append(a[0].shape[0]), real(*a, **k))[1]
    try:
        run(nw._linear)
        n_big = len(calls)
        x = x0[:100].clone().requires_grad_(True)
        nw._linear(l1, x).sum().backward()
        assert len(calls) == n_big and n_big >= 4
    finally:
        nw._fused.linear_weight_grad = real
    with torch.no_grad():
        assert nw._linear(l1, x0).grad_fn is None
    # ... and in eval mode (the inference loop differentiates the SDF network w.r.t. positions only: no weight gradient to speed up)
    l1.eval()
    before = len(calls)
    x = x0.clone().requires_grad_(True)
    nw._fused.linear_weight_grad = lambda *a, **k: (calls.append(a[0].shape[0]), real(*a, **k))[1]
    try:
        nw._linear(l1, x).sum().backward()
    finally:
        nw._fused.linear_weight_grad = real
    assert len(calls) == before


def test_accumulate_and_empty_batches():
    """accumulate != 0 adds to dW / db; an empty batch leaves them (accumulate) or zeroes them; the workspace size is what the call checks"""
    import ctypes
    import torch
    from envidr_amd import _lib
    from envidr_amd.fused import _bind_render
    lib = _lib.load()
    _bind_render(lib)
    M, K, N = 5000, 24, 32
    x, gy = torch.randn(M, K, device="cuda"), torch.randn(M, N, device="cuda")
    dW, db = torch.full((N, K), 2.0, device="cuda"), torch.full((N,), -1.0, device="cuda")
    nbytes = int(lib.envidr_linear_weight_grad_workspace_bytes(M, K, N))
    ws = torch.empty(nbytes // 4, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    assert lib.envidr_linear_weight_grad(x.data_ptr(), gy.data_ptr(), M, K, N, dW.data_ptr(), db.data_ptr(), 1, ws.data_ptr(), nbytes, s) == 0
    torch.cuda.synchronize()
    assert torch.allclose(dW, 2.0 + gy.t() @ x, atol=2e-4) and torch.allclose(db, -1.0 + gy.sum(0), atol=2e-4)
    before = dW.clone()
    assert lib.envidr_linear_weight_grad(None, None, 0, K, N, dW.data_ptr(), db.data_ptr(), 1, None, 0, s) == 0          # empty, accumulate: untouched
    torch.cuda.synchronize()
    assert torch.equal(dW, before)
    assert lib.envidr_linear_weight_grad(None, None, 0, K, N, dW.data_ptr(), db.data_ptr(), 0, None, 0, s) == 0          # empty, overwrite: zeros
    torch.cuda.synchronize()
    assert not dW.any() and not db.any()
    assert lib.envidr_linear_weight_grad(x.data_ptr(), gy.data_ptr(), M, K, N, dW.data_ptr(), None, 0, ws.data_ptr(), nbytes - 16, s) == -1
    assert b"workspace" in lib.envidr_last_error()
