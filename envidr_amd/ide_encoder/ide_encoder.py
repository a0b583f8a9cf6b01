"""`ide_encoder.IntegratedDirEncoder` backed by a HIP operator (the reference implements it in
PyTorch: ide_encoder/ide_encoder.py:57-130).  Same constructor, `output_dim` and forward
signature; forward-only (the render path never differentiates through it)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _lib


class IntegratedDirEncoder(nn.Module):
    def __init__(self, input_dim=3, deg_view=4):
        super().__init__()
        if deg_view > 5:
            raise ValueError("Only deg_view of at most 5 is numerically stable.")
        self.deg_view = deg_view
        self.output_dim = (2 ** deg_view - 1 + deg_view) * 2

    def forward(self, xyz, roughness=0, **kwargs):
        """xyz [..., 3]; roughness: scalar or [..., 1] (kappa^-1) -> [..., output_dim] = [Re | Im]"""
        prefix = list(xyz.shape[:-1])
        d = xyz.reshape(-1, 3).float().contiguous()
        B = d.shape[0]
        out = torch.empty(B, self.output_dim, dtype=torch.float32, device=d.device)
        if isinstance(roughness, torch.Tensor):
            r = roughness.reshape(-1).float().contiguous()
            if r.numel() == 1:
                _lib.call("ide_encode_forward", d, None, float(r.item()), B, self.deg_view, out)
            else:
                assert r.numel() == B, "roughness must be a scalar or one value per direction"
                _lib.call("ide_encode_forward", d, r, 0.0, B, self.deg_view, out)
        else:
            _lib.call("ide_encode_forward", d, None, float(roughness), B, self.deg_view, out)
        return out.reshape(prefix + [self.output_dim])
