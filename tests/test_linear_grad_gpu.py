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
    # ... and the normals pass (autograd.grad w.r.t. the positions inside hashencoder.input_gradient_only, renderer.compute_normal -- in eval
    # mode too: the inference loop's autograd normals) never forms a weight gradient: autograd.grad would throw it away
    from envidr_amd.hashencoder.hashgrid import input_gradient_only
    l1.eval()
    before = len(calls)
    x = x0.clone().requires_grad_(True)
    nw._fused.linear_weight_grad = lambda *a, **k: (calls.append(a[0].shape[0]), real(*a, **k))[1]
    try:
        y = nw._linear(l1, x)
        with input_gradient_only():
            gx, = torch.autograd.grad(y.sum(), x, create_graph=True)
    finally:
        nw._fused.linear_weight_grad = real
    assert len(calls) == before
    assert float((gx - l1.weight.sum(0)).abs().max()) <= 1e-5


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


@pytest.mark.parametrize("M,K,N,view", [(1, 4, 1, False), (130, 72, 256, False), (4097, 256, 256, False), (4097, 256, 256, True), (20000, 256, 12, False),
                                        (20000, 12, 256, True), (33333, 64, 64, False), (33333, 64, 64, True), (9999, 32, 64, False), (5000, 64, 15, False),
                                        (5000, 16, 64, True), (3000, 128, 3, False), (3000, 300, 260, False), (777, 160, 160, True), (2048, 100, 37, True),
                                        (1500, 64, 32, False), (1500, 64, 32, True), (1500, 12, 20, True), (2048, 100, 100, True), (2048, 100, 20, True)])
def test_rows_product_matches_float64(M, K, N, view):
    """envidr_linear_rows (csrc/linear_rows.hip) in its four epilogues against float64: both vector layouts of W (row-major [N, K]; the
    transposed view of a [K, N] matrix -- the input-gradient product) and the strided fallback, whole and ragged slabs / column blocks / row
    tiles, the uniform-base fast addressing (K % 16 == 0, N a whole column block) and the clamped general path.  Tolerance: 2e-6 of max |y|
    (an fp32 dot product of K <= 300 terms; torch's own fp32 GEMM lands at 2e-7 .. 8e-7 on the same inputs)."""
    import torch
    from envidr_amd import fused
    g = torch.Generator(device="cuda").manual_seed(M + 7 * K + 13 * N)
    x = torch.randn(M, K, device="cuda", generator=g)
    Wm = torch.randn((K, N) if view else (N, K), device="cuda", generator=g) / K ** 0.5
    W = Wm.t() if view else Wm
    if (M, K) == (2048, 100):
        W = torch.randn(N, 2 * K + 1, device="cuda", generator=g)[:, ::2][:, :K]       # neither index contiguous: the element-wise staging (N = 37 / 100 / 20: column blocks of 64 / 128 / 32)
    b = torch.randn(N, device="cuda", generator=g)
    act = torch.randn(M, N, device="cuda", generator=g)
    ref = x.double() @ W.double().t()
    tol = 2e-6 * float(ref.abs().max() + b.abs().max())
    assert float((fused.linear_rows(x, W).double() - ref).abs().max()) <= tol
    assert float((fused.linear_rows(x, W, bias=b).double() - (ref + b.double())).abs().max()) <= tol
    y = fused.linear_rows(x, W, bias=b, relu=True)
    assert float((y.double() - torch.relu(ref + b.double())).abs().max()) <= tol and float(y.min()) >= 0.0
    ym = fused.linear_rows(x, W, mask_act=act)
    assert float((ym.double() - ref * (act > 0)).abs().max()) <= tol
    assert torch.equal(ym == 0, (act <= 0) | (ym == 0))                        # masked entries are exact zeros
    assert torch.equal(fused.linear_rows(x, W), fused.linear_rows(x, W))       # deterministic
    # rows with a pitch (a column slice of a wider matrix) are read in place
    wide = torch.randn(M, K + 8, device="cuda", generator=g)
    assert float((fused.linear_rows(wide[:, 4:4 + K], W).double() - wide[:, 4:4 + K].double() @ W.double().t()).abs().max()) <= tol


def test_rows_product_refuses_what_it_cannot_take():
    import torch
    from envidr_amd import _lib, fused
    x = torch.randn(64, 6, device="cuda")
    assert not fused.linear_rows_supported(x, torch.randn(8, 6, device="cuda"))          # K % 4 != 0: the caller keeps the library GEMM
    with pytest.raises(_lib.EnvidrError):
        fused.linear_rows(x, torch.randn(8, 6, device="cuda"))
    with pytest.raises(_lib.EnvidrError):
        fused.linear_rows(torch.randn(64, 8, device="cuda"), torch.randn(8, 8, device="cuda"), relu=True)
    assert fused.linear_rows(torch.randn(0, 8, device="cuda"), torch.randn(8, 8, device="cuda")).shape == (0, 8)


def test_relu_mlp_node_has_the_layerwise_gradients():
    """`_run_mlp(..., first_order_only=True)` in the training branch -- one autograd node for the whole ReLU MLP, bias / ReLU / ReLU gradient in
    the products' epilogues.  Output against the same layers run by torch; gradients (input, weights, biases) against the chain rule in float64
    on the node's OWN activations (two fp32 GEMMs disagree on the sign of a handful of pre-activations within 1e-7 of zero, so torch's backward
    differs from ANY other fp32 forward by ~1e-3 of the gradient norm through those few ReLU masks; torch's gradients are checked to be that
    close).  Shapes: the environment MLP 72-256-256-256-12 and a colour head ending in 3 outputs (whose input-gradient product torch keeps)."""
    import torch
    from envidr_amd import fused
    from envidr_amd.nerf import network as nw
    for dims in ((72, 256, 256, 256, 12), (40, 64, 64, 3), (16, 32, 12)):
        torch.manual_seed(len(dims))
        net = nw._mlp(list(dims)).cuda().train()
        x0 = torch.randn(20000, dims[0], device="cuda")
        w = torch.randn(20000, dims[-1], device="cuda")

        def run(fn):
            net.zero_grad()
            x = x0.clone().requires_grad_(True)
            y = fn(net, x)
            (y * w).sum().backward()
            return [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in net.parameters()]

        def plain(net_, h):
            for i, lin in enumerate(net_):
                h = lin(h)
                if i != len(net_) - 1:
                    h = torch.relu(h)
            return h
        calls = []
        real = nw._fused.linear_rows
        nw._fused.linear_rows = lambda *a, **k: (calls.append(k), real(*a, **k))[1]
        try:
            got = run(lambda n, h: nw._run_mlp(n, h, first_order_only=True))
        finally:
            nw._fused.linear_rows = real
        want = run(plain)
        assert float((got[0] - want[0]).norm() / want[0].norm()) <= 5e-6
        for a, b in zip(got[1:], want[1:]):
            assert float((a - b).norm() / b.norm()) <= 2e-2, (dims, float((a - b).norm() / b.norm()))
        # the chain rule in float64 on the node's activations
        with torch.no_grad():
            acts, h = [x0], x0
            for i, lin in enumerate(net):
                h = nw._rows_product(h, lin.weight, bias=lin.bias, relu=i != len(net) - 1)
                acts.append(h)
            assert torch.equal(h, got[0])
            g = w.double()
            exact = []
            for i in range(len(net) - 1, -1, -1):
                exact = [g.t() @ acts[i].double(), g.sum(0)] + exact
                g = g @ net[i].weight.double()
                if i > 0:
                    g = g * (acts[i] > 0)
            exact = [g] + exact
        for a, b in zip(got[1:], exact):
            assert float((a.double() - b).norm() / b.norm()) <= 5e-6, (dims, float((a.double() - b).norm() / b.norm()))
        assert sum(1 for k in calls if k.get("relu")) == len(dims) - 2                      # every hidden layer: bias + ReLU in the epilogue
        assert sum(1 for k in calls if k.get("mask_act") is not None) >= len(dims) - 3      # the ReLU gradient rides on the gradient products
        # without the flag (a network that is differentiated twice) or below the row threshold the per-layer path runs
        y = nw._run_mlp(net, x0[:100].clone().requires_grad_(True), first_order_only=True)
        assert "ReluMlp" not in type(y.grad_fn).__name__


def test_row_expanded_gradients_reach_the_operator_as_copies():
    """the gradient of y.sum(0) (or y.mean(0)) arrives in a backward as an EXPANDED tensor, strides (0, 1): rows that overlap.  The operator
    wants ldx >= K; linear_rows copies such an operand instead of refusing it (round-5 advisor finding)"""
    import torch
    from envidr_amd.fused import linear_rows
    from envidr_amd.nerf import network as nw
    torch.manual_seed(1)
    W = torch.randn(48, 64, device="cuda")
    row = torch.randn(1, 64, device="cuda")
    x = row.expand(300, 64)
    assert x.stride() == (0, 1)
    y = linear_rows(x, W)
    want = (x.double() @ W.double().t())
    assert float((y.double() - want).abs().max()) <= 1e-4
    mask = torch.randn(1, 48, device="cuda").expand(300, 48)
    ym = linear_rows(x, W, mask_act=mask)
    assert float((ym.double() - want * (mask > 0)).abs().max()) <= 1e-4
    # ... and through the autograd nodes of the training branch: loss = y.sum(0) makes every incoming gradient row-expanded
    lin = torch.nn.Linear(64, 48).cuda().train()
    net = torch.nn.ModuleList([torch.nn.Linear(64, 64), torch.nn.Linear(64, 12)]).cuda().train()
    h = torch.randn(nw.WEIGHT_GRAD_OPERATOR_MIN_ROWS + 5, 64, device="cuda", requires_grad=True)
    for fn, params in ((lambda t: nw._linear(lin, t), list(lin.parameters())), (lambda t: nw._run_mlp(net, t, first_order_only=True), list(net.parameters()))):
        got = torch.autograd.grad(fn(h).sum(0).sum(), [h] + params)
        ref_fn = (lambda t: lin(t)) if params[0] is lin.weight else (lambda t: net[1](torch.relu(net[0](t))))
        want_g = torch.autograd.grad(ref_fn(h).sum(0).sum(), [h] + params)
        for a, b in zip(got, want_g):
            assert float((a - b).abs().max()) <= 2e-3 * max(1.0, float(b.abs().max()))
