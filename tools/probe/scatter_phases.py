"""Block-level phases of k_table_scatter_lds (variant scatter_phases: five 64-bit words behind the table gradient)"""
import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from envidr_amd import _lib, scenes
dev = torch.device("cuda:0")
sc = scenes.toaster_scene()
offsets = torch.from_numpy(np.ascontiguousarray(sc.offsets, np.int32)).to(dev)
S = float(np.log2(sc.per_level_scale))
M, L = 7_700_000, 16
x01 = torch.rand(M, 3, device=dev)
grad = torch.randn(L, M, 2, device=dev)
rows = int(sc.offsets[L])
gtab = torch.zeros(rows + 8, 2, device=dev)
for i in range(2):
    gtab.zero_()
    _lib.call("hash_encode_backward", grad, x01, gtab, offsets, gtab, M, 3, 2, L, S, 16, 0, None, None)
torch.cuda.synchronize()
t = gtab[rows:].contiguous().view(torch.int64).reshape(-1).cpu().numpy()[:5] * 64.0
names = ["zeroing + barrier", "point loop (barrier to barrier)", "flush", "whole kernel (mean over blocks)", "whole kernel (max over blocks)"]
for n, v, d in zip(names, t, (256, 256, 256, 256, 1)): print(f"{n:36s} {float(v) / d / 1e6:8.2f} M cycles per workgroup")
