"""Ray generation and image metrics used around the render path (the reference keeps them in
nerf/utils.py next to its Trainer, which is out of scope here)."""
from __future__ import annotations

import math

import numpy as np
import torch


def rot_theta(th: float) -> np.ndarray:
    """4x4 rotation about the y axis, the convention env rotation uses (reference nerf/utils.py:48-52)"""
    c, s = math.cos(th), math.sin(th)
    return np.array([[c, 0, -s, 0], [0, 1, 0, 0], [s, 0, c, 0], [0, 0, 0, 1]], dtype=np.float64)


@torch.no_grad()
def get_rays(poses: torch.Tensor, intrinsics, H: int, W: int, N: int = -1) -> dict:
    """poses [B,4,4] camera-to-world, intrinsics (fx, fy, cx, cy) -> rays_o, rays_d [B, n, 3]
    (pixel centres at +0.5, unit directions; N > 0 draws N random pixels like the reference's
    training sampler, N <= 0 returns the full image in row-major order; reference :109-209)."""
    device, B = poses.device, poses.shape[0]
    fx, fy, cx, cy = intrinsics
    ys, xs = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                            torch.arange(W, device=device, dtype=torch.float32), indexing="ij")
    i = xs.reshape(1, H * W).expand(B, H * W) + 0.5
    j = ys.reshape(1, H * W).expand(B, H * W) + 0.5
    out = {}
    if N > 0:
        inds = torch.randint(0, H * W, size=[min(N, H * W)], device=device).expand(B, -1)
        i, j = torch.gather(i, -1, inds), torch.gather(j, -1, inds)
        out["inds"] = inds
    dirs = torch.stack(((i - cx) / fx, (j - cy) / fy, torch.ones_like(i)), dim=-1)
    dirs = dirs / torch.norm(dirs, dim=-1, keepdim=True)
    out["rays_d"] = dirs @ poses[:, :3, :3].transpose(-1, -2)
    out["rays_o"] = poses[..., :3, 3][..., None, :].expand_as(out["rays_d"])
    return out


class PSNRMeter:
    """running PSNR = -10 log10(mean((pred - truth)^2)) per update, averaged (reference :278-312)"""

    def __init__(self):
        self.V, self.N = 0.0, 0

    def clear(self):
        self.V, self.N = 0.0, 0

    def update(self, preds, truths):
        p = preds.detach().cpu().numpy() if torch.is_tensor(preds) else np.asarray(preds)
        t = truths.detach().cpu().numpy() if torch.is_tensor(truths) else np.asarray(truths)
        self.V += -10 * np.log10(np.mean((p.astype(np.float64) - t) ** 2))
        self.N += 1

    def measure(self):
        return self.V / max(self.N, 1)
