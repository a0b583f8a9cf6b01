"""`ide_encoder.IntegratedDirEncoder` backed by HIP operators (the reference implements it in
PyTorch: ide_encoder/ide_encoder.py:57-130).  Same constructor, `output_dim` and forward
signature.  Differentiable w.r.t. the direction and the roughness like the reference's torch
formulation (its training branch back-propagates colours through the encoding into the normals and
the roughness head): `envidr_ide_encode_backward`."""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _lib


class _IdeEncode(Function):
    """forward / backward on the HIP operators; `rough` is a [B] tensor (per direction), a 1-element tensor (shared) or None
    with the shared value in `scalar`"""

    @staticmethod
    def forward(ctx, d, rough, scalar, deg):
        B = d.shape[0]
        out = torch.empty(B, (2 ** deg - 1 + deg) * 2, dtype=torch.float32, device=d.device)
        shared = rough is None or rough.numel() == 1
        if rough is not None and shared:
            scalar = float(rough.reshape(-1)[0].item())
        _lib.call("ide_encode_forward", d, None if shared else rough, float(scalar), B, deg, out)
        ctx.save_for_backward(d, rough if rough is not None else torch.empty(0, device=d.device))
        ctx.meta = (B, deg, float(scalar), shared, rough is not None)
        return out

    @staticmethod
    def backward(ctx, grad):
        d, rough = ctx.saved_tensors
        B, deg, scalar, shared, has_rough = ctx.meta
        need_d, need_r = ctx.needs_input_grad[0], has_rough and ctx.needs_input_grad[1]
        if not (need_d or need_r) or B == 0:
            return None, None, None, None
        gd = torch.empty(B, 3, dtype=torch.float32, device=d.device) if need_d else None
        gr = torch.empty(B, dtype=torch.float32, device=d.device) if need_r else None
        _lib.call("ide_encode_backward", grad.float().contiguous(), d, None if shared else rough, scalar, B, deg, gd, gr)
        if need_r:
            gr = gr.sum().reshape(rough.shape) if shared else gr.reshape(rough.shape)
        return gd, gr, None, None


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
        if isinstance(roughness, torch.Tensor):
            r = roughness.reshape(-1).float().contiguous()
            if r.numel() != 1 and r.numel() != B:
                raise _lib.EnvidrError("roughness must be a scalar or one value per direction")
            out = _IdeEncode.apply(d, r, 0.0, self.deg_view)
        else:
            out = _IdeEncode.apply(d, None, float(roughness), self.deg_view)
        return out.reshape(prefix + [self.output_dim])
