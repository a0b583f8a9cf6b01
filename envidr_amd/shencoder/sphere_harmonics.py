"""`shencoder`: real spherical harmonics up to degree 8 on the HIP library (mirrors
shencoder/sphere_harmonics.py:14-87)."""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _lib


class _SHEncode(Function):
    @staticmethod
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        inputs = inputs.float().contiguous()
        B, D = inputs.shape
        outputs = torch.empty(B, degree ** 2, dtype=torch.float32, device=inputs.device)
        dy_dx = torch.empty(B, D * degree ** 2, dtype=torch.float32, device=inputs.device) if calc_grad_inputs else None
        _lib.call("sh_encode_forward", inputs, outputs, B, D, degree, dy_dx)
        ctx.save_for_backward(inputs, dy_dx if dy_dx is not None else torch.empty(1, device=inputs.device))
        ctx.dims = (B, D, degree)
        ctx.calc_grad_inputs = calc_grad_inputs
        return outputs

    @staticmethod
    def backward(ctx, grad):
        if not ctx.calc_grad_inputs:
            return None, None, None
        inputs, dy_dx = ctx.saved_tensors
        B, D, degree = ctx.dims
        grad_inputs = torch.zeros_like(inputs)
        _lib.call("sh_encode_backward", grad.contiguous(), inputs, B, D, degree, dy_dx, grad_inputs)
        return grad_inputs, None, None


sh_encode = _SHEncode.apply


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim, self.degree = input_dim, degree
        self.output_dim = degree ** 2
        assert self.input_dim == 3, "SH encoder only support input dim == 3"
        assert 0 < self.degree <= 8, "SH encoder only supports degree in [1, 8]"

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        inputs = inputs / size
        prefix = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        return sh_encode(inputs, self.degree, inputs.requires_grad).reshape(prefix + [self.output_dim])
