"""`hashencoder` (encoding_pos = hashgrid_diff): smoothstep multiresolution hash grid with analytic
input derivative and first/second-order backward, on the HIP library.  Mirrors the reference's
hashencoder/hashgrid.py (hash_encode :107, HashEncoder :110-169).  fp32, or -- like the reference's
`custom_fwd(cast_inputs=torch.half)` (hashgrid.py:19) -- fp16 for inputs, table, outputs and dy_dx when the table is given in
half or torch autocast is on (`hash_encode_*_f16`, the at::Half instantiations, second backward included)."""
from __future__ import annotations

import contextlib

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _lib


class _HashEncode(Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False):
        half = embeddings.dtype == torch.half or torch.is_autocast_enabled()
        dtype = torch.half if half else torch.float32
        inputs, embeddings, offsets = inputs.to(dtype).contiguous(), embeddings.to(dtype).contiguous(), offsets.contiguous()
        B, D = inputs.shape
        L, C = offsets.shape[0] - 1, embeddings.shape[1]
        S, H = float(np.log2(per_level_scale)), int(base_resolution)
        outputs = torch.empty(L, B, C, device=inputs.device, dtype=dtype)     # level-major, like the reference
        dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=dtype) if calc_grad_inputs else None
        _lib.call("hash_encode_forward_f16" if half else "hash_encode_forward", inputs, embeddings, offsets, outputs, B, D, C, L, S, H,
                  int(calc_grad_inputs), dy_dx)
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx if dy_dx is not None else torch.empty(1, device=inputs.device, dtype=dtype))
        ctx.dims = (B, D, C, L, S, H)
        ctx.calc_grad_inputs, ctx.half = calc_grad_inputs, half
        return outputs.permute(1, 0, 2).reshape(B, L * C)

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H = ctx.dims
        grad = grad.to(embeddings.dtype).view(B, L, C).permute(1, 0, 2).contiguous()
        # whether the table gradient is wanted is known HERE (needs_input_grad); inside _HashEncodeBackward.forward grad
        # mode is always off, so asking torch.is_grad_enabled() there would never produce it
        need_table = bool(ctx.needs_input_grad[1]) and not getattr(_INPUT_GRADIENT_ONLY, "on", False)
        grad_inputs, grad_embeddings = _HashEncodeBackward.apply(grad, inputs, embeddings, offsets, B, D, C, L, S, H,
                                                                 ctx.calc_grad_inputs, dy_dx, need_table)
        return (grad_inputs if ctx.calc_grad_inputs else None), grad_embeddings, None, None, None, None


class _Flag:                       # process-wide on purpose: autograd runs the backward of GPU nodes on its own worker thread,
    on = False                     # where a thread-local set by the caller would not be seen


_INPUT_GRADIENT_ONLY = _Flag()


@contextlib.contextmanager
def input_gradient_only():
    """Inside this context a backward pass through the hash encoding computes the gradient w.r.t. the POSITIONS only.  For callers that
    run `torch.autograd.grad(features-derived scalar, positions, create_graph=True)` -- the normals of the training branch
    (renderer.py:182-185): autograd.grad returns the position gradient and throws the table gradient of that pass away, but the custom
    Function cannot know (needs_input_grad follows requires_grad), so it would zero-fill and scatter a 48.8 MB table gradient per step for
    nothing.  The pass stays differentiable: the second backward (hashencoder.cu:375-595) still delivers grad2_embeddings."""
    prev = getattr(_INPUT_GRADIENT_ONLY, "on", False)
    _INPUT_GRADIENT_ONLY.on = True
    try:
        yield
    finally:
        _INPUT_GRADIENT_ONLY.on = prev


class _HashEncodeBackward(Function):
    """first backward as a Function so that autograd can differentiate it again (eikonal loss)"""

    @staticmethod
    def forward(ctx, grad, inputs, embeddings, offsets, B, D, C, L, S, H, calc_grad_inputs, dy_dx, need_table=True):
        grad_inputs = torch.zeros_like(inputs)
        # the table gradient is only materialised when somebody consumes it (need_table = needs_input_grad of the
        # embeddings): at inference (normals only) the reference still zero-fills and scatters into 48.8 MB every iteration
        grad_embeddings = torch.zeros_like(embeddings) if need_table else None
        half = embeddings.dtype == torch.half               # the at::Half instantiations (hashencoder.cu:778,817)
        _lib.call("hash_encode_backward_f16" if half else "hash_encode_backward", grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H,
                  int(calc_grad_inputs), dy_dx if calc_grad_inputs else None, grad_inputs if calc_grad_inputs else None)
        ctx.save_for_backward(grad, inputs, embeddings, offsets, dy_dx)
        ctx.dims = (B, D, C, L, S, H)
        ctx.calc_grad_inputs = calc_grad_inputs
        return grad_inputs, grad_embeddings

    @staticmethod
    def backward(ctx, grad_grad_inputs, grad_grad_embeddings):
        grad, inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H = ctx.dims
        # The second-order kernel (reference hashencoder.cu:375-595) propagates through grad_inputs only; the reference's
        # Function ignores grad_grad_embeddings the same way (hashgrid.py:91-107).  Without grad_inputs there is nothing it can
        # do: the saved dy_dx is then the 1-element dummy of the forward pass, which the reference's kernel would read as
        # [B, L*D*C] (undefined behaviour there; refused here).
        if grad_grad_inputs is None:
            return (None,) * 13
        if not ctx.calc_grad_inputs:
            raise RuntimeError("double backward through the hash encoding needs the input gradient: the forward pass ran with "
                               "inputs.requires_grad == False, so no dy_dx was kept")
        grad_grad = torch.zeros_like(grad)
        grad2_embeddings = torch.zeros_like(embeddings)
        _lib.call("hash_encode_second_backward_f16" if embeddings.dtype == torch.half else "hash_encode_second_backward", grad, inputs, embeddings,
                  offsets, B, D, C, L, S, H, int(ctx.calc_grad_inputs), dy_dx, grad_grad_inputs.to(embeddings.dtype).contiguous(), grad_grad, grad2_embeddings)
        return grad_grad, None, grad2_embeddings, None, None, None, None, None, None, None, None, None, None


hash_encode = _HashEncode.apply


class HashEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None):
        super().__init__()
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale, self.log2_hashmap_size, self.base_resolution = per_level_scale, log2_hashmap_size, base_resolution
        self.output_dim = num_levels * level_dim
        self.max_params = 2 ** log2_hashmap_size
        offsets, offset = [], 0
        for i in range(num_levels):
            resolution = int(np.ceil(base_resolution * per_level_scale ** i))
            offsets.append(offset)
            offset += min(self.max_params, resolution ** input_dim)
        offsets.append(offset)
        self.register_buffer("offsets", torch.from_numpy(np.array(offsets, dtype=np.int32)))
        self.n_params = offsets[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(offset, level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)

    def __repr__(self):
        return (f"HashEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"base_resolution={self.base_resolution} per_level_scale={self.per_level_scale} params={tuple(self.embeddings.shape)}")

    def forward(self, inputs, bound=1, table_grad=True):
        """`table_grad=False` (not a reference argument): the table takes no part in the graph -- for callers that differentiate the
        features w.r.t. the POSITIONS only (normals at inference): autograd hands a custom Function `needs_input_grad` from the inputs'
        `requires_grad`, not from what `autograd.grad(..., inputs=xyz)` asked for, so with the Parameter in the graph every backward also
        zero-fills and scatters a 48.8 MB table gradient nobody reads (the reference does: SURVEY.md 8 a6 "wastefully")."""
        inputs = (inputs + bound) / (2 * bound)
        prefix = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        table = self.embeddings if table_grad else self.embeddings.detach()
        out = hash_encode(inputs, table, self.offsets, self.per_level_scale, self.base_resolution, inputs.requires_grad)
        return out.view(prefix + [self.output_dim])
