import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from envidr_amd import _lib, scenes
dev = torch.device("cuda:0")
sc = scenes.toaster_scene()
offsets = torch.from_numpy(np.ascontiguousarray(sc.offsets, np.int32)).to(dev)
S = float(np.log2(sc.per_level_scale))
M = 7_700_000
x01 = torch.rand(M, 3, device=dev)
def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for L in (1, 2, 3, 4, 5, 6, 8, 16):
    grad = torch.randn(L, M, 2, device=dev)
    gtab = torch.zeros(int(sc.offsets[L]), 2, device=dev)
    ms = t(lambda: _lib.call("hash_encode_backward", grad, x01, gtab, offsets[:L + 1].contiguous(), gtab, M, 3, 2, L, S, 16, 0, None, None))
    print(f"levels 0..{L-1}: {ms:8.2f} ms")
