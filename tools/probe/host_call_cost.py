"""HOST time per C-ABI call (enqueue only, no synchronisation inside the loop) for the operators of a training step."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from envidr_amd import _lib, fused, scenes
dev = torch.device("cuda:0")
sc = scenes.toaster_scene()
table = torch.from_numpy(sc.table).to(dev)
offsets = torch.from_numpy(np.ascontiguousarray(sc.offsets, np.int32)).to(dev)
S = float(np.log2(sc.per_level_scale))
M = 145920
x01 = torch.rand(M, 3, device=dev); out = torch.empty(16, M, 2, device=dev); dy = torch.empty(M, 96, device=dev)
grad = torch.randn(16, M, 2, device=dev); gin = torch.zeros(M, 3, device=dev); gtab = torch.zeros_like(table)
ggx = torch.randn(M, 3, device=dev); gg = torch.zeros(16, M, 2, device=dev)
xa = torch.randn(M, 64, device=dev); W = torch.randn(64, 64, device=dev); b = torch.randn(64, device=dev)
def host(fn, n=100):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6
rows = [
 ("hash_encode_forward (+dy_dx)", lambda: _lib.call("hash_encode_forward", x01, table, offsets, out, M, 3, 2, 16, S, 16, 1, dy)),
 ("hash_encode_backward (inputs only)", lambda: _lib.call("hash_encode_backward", grad, x01, table, offsets, None, M, 3, 2, 16, S, 16, 1, dy, gin)),
 ("hash_encode_backward (+ table)", lambda: _lib.call("hash_encode_backward", grad, x01, table, offsets, gtab, M, 3, 2, 16, S, 16, 1, dy, gin)),
 ("hash_encode_second_backward", lambda: _lib.call("hash_encode_second_backward", grad, x01, table, offsets, M, 3, 2, 16, S, 16, 1, dy, ggx, gg, gtab)),
 ("fused.linear_rows 64x64", lambda: fused.linear_rows(xa, W, bias=b, relu=True)),
 ("fused.linear_weight_grad 64x64", lambda: fused.linear_weight_grad(xa, xa)),
 ("torch addmm 64x64", lambda: torch.addmm(b, xa, W.t())),
 ("torch relu", lambda: torch.relu(xa)),
 ("torch.empty", lambda: torch.empty(M, 64, device=dev)),
]
for name, fn in rows:
    print(f"{name:40s} {host(fn):7.1f} us of host time per call")
