"""`trunc_exp` (reference activation.py:5-19): exp in the forward pass, and in the backward pass the exponential of the input CLAMPED to
[-15, 15] -- the density activation of the plain-NeRF branch (`use_sdf = False`, network.py:424-429), which keeps a gradient alive where
exp has overflowed or underflowed.  Built from differentiable torch operations on the saved input, so it can be differentiated again
(the normals are taken with create_graph)."""
from __future__ import annotations

import torch


class _TruncExp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.float()                                   # (the reference casts to float32 under autocast: custom_fwd(cast_inputs=torch.float32))
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply
