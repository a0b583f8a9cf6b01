import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from envidr_amd import _lib, scenes
dev = torch.device("cuda:0")
sc = scenes.toaster_scene()
offsets = torch.from_numpy(np.ascontiguousarray(sc.offsets, np.int32)).to(dev)
S = float(np.log2(sc.per_level_scale))
M = 7_700_000
x01 = torch.rand(M, 3, device=dev)
L = 16
grad = torch.randn(L, M, 2, device=dev)
gtab = torch.zeros(int(sc.offsets[L]), 2, device=dev)
for _ in range(3):
    _lib.call("hash_encode_backward", grad, x01, gtab, offsets, gtab, M, 3, 2, L, S, 16, 0, None, None)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    _lib.call("hash_encode_backward", grad, x01, gtab, offsets, gtab, M, 3, 2, L, S, 16, 0, None, None)
torch.cuda.synchronize()
print(f"hash_encode_backward + table scatter, {M} points: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
