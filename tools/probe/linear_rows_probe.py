"""envidr_linear_rows against torch (hipBLASLt / rocBLAS) on the shapes of a training step: value check against float64, time per call."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from envidr_amd import fused

dev = torch.device("cuda:0")
torch.manual_seed(0)
def timeit(fn, reps=20, batches=3):
    """ms per call: the fastest of `batches` event-timed loops (a stall of the shared host inside one loop would otherwise be the figure)"""
    fn(); torch.cuda.synchronize()
    best = float("inf")
    for _ in range(batches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best
M = int(sys.argv[1]) if len(sys.argv) > 1 else 145920
for K, N, trans in ((256, 256, False), (256, 256, True), (72, 256, False), (256, 72, True), (256, 12, False), (12, 256, True), (64, 64, False), (64, 64, True),
                    (32, 64, False), (64, 16, False), (16, 64, True), (128, 3, False), (160, 160, False), (300, 260, False)):
    x = torch.randn(M, K, device=dev)
    Wm = torch.randn(K, N, device=dev) / K ** 0.5 if trans else torch.randn(N, K, device=dev) / K ** 0.5       # trans: the stored matrix is [K, N], used as W = Wm.t()
    W = Wm.t() if trans else Wm
    b = torch.randn(N, device=dev)
    ref = (x.double() @ W.double().t())
    y0 = fused.linear_rows(x, W)
    y1 = fused.linear_rows(x, W, bias=b)
    y2 = fused.linear_rows(x, W, bias=b, relu=True)
    act = torch.randn(M, N, device=dev)
    y3 = fused.linear_rows(x, W, mask_act=act)
    scale = ref.abs().max().item()
    errs = [(y0.double() - ref).abs().max().item() / scale, (y1.double() - (ref + b.double())).abs().max().item() / scale,
            (y2.double() - torch.relu(ref + b.double())).abs().max().item() / scale, (y3.double() - ref * (act > 0)).abs().max().item() / scale]
    terr = ((x @ W.t()).double() - ref).abs().max().item() / scale
    t_ours = timeit(lambda: fused.linear_rows(x, W, bias=b, relu=True))
    t_plain = timeit(lambda: fused.linear_rows(x, W))
    Wc = W.contiguous()
    t_torch = timeit(lambda: torch.relu(torch.addmm(b, x, Wc.t())))
    t_mm = timeit(lambda: x @ Wc.t())
    fl = 2.0 * M * K * N
    print(f"M {M} K {K} N {N} {'W^T view' if trans else 'W [N,K]'}: max err / max|y| plain {errs[0]:.1e} bias {errs[1]:.1e} relu {errs[2]:.1e} mask {errs[3]:.1e} (torch fp32: {terr:.1e}) | "
          f"ours bias+relu {t_ours * 1e3:.1f} us ({fl / t_ours / 1e9:.1f} TFLOP/s), plain {t_plain * 1e3:.1f} us | torch addmm+relu {t_torch * 1e3:.1f} us, mm {t_mm * 1e3:.1f} us ({fl / t_mm / 1e9:.1f} TFLOP/s)")
