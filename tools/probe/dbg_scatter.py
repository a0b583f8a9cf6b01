import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tests.test_scatter_gpu import _setup
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
    gg = torch.zeros(L, n, C, device=dev)
    _lib.call("hash_encode_second_backward", grad[:, lo:hi].contiguous(), x[lo:hi].contiguous(), table, offsets, n, D, C, L, S, H, 1, dy[lo:hi].contiguous(), ggx[lo:hi].contiguous(), gg, into)
big = torch.zeros(rows, C, device=dev); run(0, B, big)
small = torch.zeros(rows, C, device=dev)
third = B // 3
for lo, hi in ((0, third), (third, 2 * third), (2 * third, B)): run(lo, hi, small)
big2 = torch.zeros(rows, C, device=dev); run(0, B, big2)
a, b, a2 = big.cpu().numpy().astype(np.float64), small.cpu().numpy().astype(np.float64), big2.cpu().numpy().astype(np.float64)
mis = (a != 0) != (b != 0)
print("pattern mismatches", mis.sum(), "a nonzero", (a != 0).sum(), "b nonzero", (b != 0).sum(), "run-to-run pattern diff", ((a != 0) != (a2 != 0)).sum())
idx = np.argwhere(mis)[:10]
for r, c in idx:
    lv = np.searchsorted(sc.offsets, r, side="right") - 1
    print("row", r, "level", lv, "a", a[r, c], "b", b[r, c])
for l in range(L):
    s = slice(int(sc.offsets[l]), int(sc.offsets[l + 1]))
    print(l, np.abs(a[s] - b[s]).max() / (np.abs(b[s]).max() + 1e-30), np.abs(a[s]-a2[s]).max() / (np.abs(b[s]).max() + 1e-30))
