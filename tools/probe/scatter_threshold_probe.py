"""the table scatter of hash_encode_backward either side of kLdsScatterMinPoints (2^16): per-point atomics below, LDS ranges above"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from envidr_amd import _lib, scenes
dev = torch.device("cuda:0")
sc = scenes.toaster_scene()
table = torch.from_numpy(sc.table).to(dev)
offsets = torch.from_numpy(np.ascontiguousarray(sc.offsets, np.int32)).to(dev)
S = float(np.log2(sc.per_level_scale))
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
gtab = torch.zeros_like(table)
for M in (16384, 32768, 49152, 65535, 65536, 81920, 98304, 145920, 262144, 524287, 524288):
    # points along short segments (consecutive samples of a ray), like a training batch
    n_rays = M // 32
    o = torch.rand(n_rays, 1, 3, device=dev) * 0.6 + 0.2
    d = torch.nn.functional.normalize(torch.randn(n_rays, 1, 3, device=dev), dim=-1)
    x = (o + d * torch.arange(32, device=dev).view(1, 32, 1) * 0.0034).reshape(-1, 3).clamp(0, 1).contiguous()
    x = torch.cat([x, torch.rand(M - x.shape[0], 3, device=dev)]) if x.shape[0] < M else x
    grad = torch.randn(16, M, 2, device=dev)
    t = timeit(lambda: _lib.call("hash_encode_backward", grad, x, table, offsets, gtab, M, 3, 2, 16, S, 16, 0, None, None))
    print(f"{M:7d} points: {t * 1e3:7.1f} us  ({M * 256 / t / 1e6:.1f} G atomics/s)")
