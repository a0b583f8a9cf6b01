"""geometry stage time against the ORDER the rays of a frame are listed in (blocks of 64 consecutive rays share a wave; the
sample list follows the ray order): row-major pixels, 8x8 tiles row-major, 8x8 tiles in Z-order, 16x4 and 4x16 tiles"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from envidr_amd import scenes
from envidr_amd.fused import FusedRenderer
dev = torch.device("cuda:0")
r = FusedRenderer.from_scene(scenes.toaster_scene(), device=dev)
H = W = 800
o, d = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(H, W))

def tiles(th, tw, zorder):
    ty, tx = H // th, W // tw
    yy, xx = np.meshgrid(np.arange(ty), np.arange(tx), indexing="ij")
    if zorder:
        def part(v):
            v = v.astype(np.uint32); v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555; return v
        key = (part(yy.ravel()) << 1) | part(xx.ravel())
        order = np.argsort(key, kind="stable")
    else:
        order = np.arange(ty * tx)
    y0, x0 = yy.ravel()[order] * th, xx.ravel()[order] * tw
    dy, dx = np.meshgrid(np.arange(th), np.arange(tw), indexing="ij")
    return ((y0[:, None] + dy.ravel()[None]) * W + x0[:, None] + dx.ravel()[None]).ravel()

orders = {"row-major": np.arange(H * W), "8x8 tiles": tiles(8, 8, False), "8x8 tiles, Z-order": tiles(8, 8, True), "4x16 tiles": tiles(4, 16, False),
          "16x4 tiles": tiles(16, 4, False), "4x16 tiles, Z-order": tiles(4, 16, True), "2x32 tiles": tiles(2, 32, False)}
for name, idx in orders.items():
    assert np.array_equal(np.sort(idx), np.arange(H * W))
    ix = torch.from_numpy(idx).to(dev)
    oo, dd = o[ix].contiguous(), d[ix].contiguous()
    for hint in (True, False):
        out = {}
        for _ in range(3):
            r.render_frame(oo, dd, 0.1, out=out, geometry_only=True, use_cost_hint=hint)
        ms = []
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r.render_frame(oo, dd, 0.1, out=out, geometry_only=True, use_cost_hint=hint, wait=False); e1.record()
            torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
        print(f"{name:22s} hint={hint!s:5s} geometry {np.median(ms):.3f} ms")
