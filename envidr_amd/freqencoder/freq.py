"""`freqencoder`: NeRF positional encoding on the HIP library (mirrors freqencoder/freq.py:15-77)."""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _lib


class _FreqEncode(Function):
    @staticmethod
    def forward(ctx, inputs, degree, output_dim):
        inputs = (inputs if inputs.is_cuda else inputs.cuda()).float().contiguous()
        B, D = inputs.shape
        outputs = torch.empty(B, output_dim, dtype=torch.float32, device=inputs.device)
        _lib.call("freq_encode_forward", inputs, B, D, degree, output_dim, outputs)
        ctx.save_for_backward(inputs, outputs)
        ctx.dims = (B, D, degree, output_dim)
        return outputs

    @staticmethod
    def backward(ctx, grad):
        inputs, outputs = ctx.saved_tensors
        B, D, degree, C = ctx.dims
        grad_inputs = torch.zeros_like(inputs)
        _lib.call("freq_encode_backward", grad.contiguous(), outputs, B, D, degree, C, grad_inputs)
        return grad_inputs, None, None


freq_encode = _FreqEncode.apply


class FreqEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim, self.degree = input_dim, degree
        self.output_dim = input_dim + input_dim * 2 * degree

    def __repr__(self):
        return f"FreqEncoder: input_dim={self.input_dim} degree={self.degree} output_dim={self.output_dim}"

    def forward(self, inputs, **kwargs):
        prefix = list(inputs.shape[:-1])
        out = freq_encode(inputs.reshape(-1, self.input_dim), self.degree, self.output_dim)
        return out.reshape(prefix + [self.output_dim])
