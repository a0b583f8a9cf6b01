"""`gridencoder` (encoding = hashgrid / tiledgrid): torch-ngp's linear-interpolation grid on the HIP
library.  Mirrors gridencoder/grid.py (grid_encode :89, GridEncoder :92-175).  The table may be fp32 or fp16: like the
reference (grid.py:37-40) the inputs stay fp32 and a half table -- given as such, or narrowed under torch autocast when the
channel count is even -- selects the at::Half instantiation (`grid_encode_*_f16`); outputs / dy_dx take the table's dtype."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _lib

_gridtype_to_id = {"hash": 0, "tiled": 1}


class _GridEncode(Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False):
        inputs, offsets = inputs.float().contiguous(), offsets.contiguous()
        B, D = inputs.shape
        L, C = offsets.shape[0] - 1, embeddings.shape[1]
        S, H = float(np.log2(per_level_scale)), int(base_resolution)
        # manual autocast like the reference: only the embeddings go to half, and only for even C (grid.py:37-40)
        if torch.is_autocast_enabled() and C % 2 == 0:
            embeddings = embeddings.to(torch.half)
        half = embeddings.dtype == torch.half
        embeddings = embeddings.contiguous() if half else embeddings.float().contiguous()
        dtype = embeddings.dtype
        outputs = torch.empty(L, B, C, device=inputs.device, dtype=dtype)
        dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=dtype) if calc_grad_inputs else None
        _lib.call("grid_encode_forward_f16" if half else "grid_encode_forward", inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx,
                  gridtype, int(align_corners))
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx if dy_dx is not None else torch.empty(1, device=inputs.device))
        ctx.dims = (B, D, C, L, S, H, gridtype)
        ctx.calc_grad_inputs, ctx.align_corners, ctx.half = calc_grad_inputs, align_corners, half
        return outputs.permute(1, 0, 2).reshape(B, L * C)

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype = ctx.dims
        grad = grad.to(embeddings.dtype).view(B, L, C).permute(1, 0, 2).contiguous()
        grad_embeddings = torch.zeros_like(embeddings)
        grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype) if ctx.calc_grad_inputs else None
        _lib.call("grid_encode_backward_f16" if ctx.half else "grid_encode_backward", grad, inputs, embeddings, offsets, grad_embeddings, B, D, C,
                  L, S, H, dy_dx if ctx.calc_grad_inputs else None, grad_inputs, gridtype, int(ctx.align_corners))
        if grad_inputs is not None:
            grad_inputs = grad_inputs.to(inputs.dtype)
        return grad_inputs, grad_embeddings, None, None, None, None, None, None


grid_encode = _GridEncode.apply


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype="hash", align_corners=False):
        super().__init__()
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale, self.log2_hashmap_size, self.base_resolution = per_level_scale, log2_hashmap_size, base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype, self.gridtype_id, self.align_corners = gridtype, _gridtype_to_id[gridtype], align_corners
        self.max_params = 2 ** log2_hashmap_size
        offsets, offset = [], 0
        for i in range(num_levels):
            resolution = int(np.ceil(base_resolution * per_level_scale ** i))
            n = min(self.max_params, (resolution if align_corners else resolution + 1) ** input_dim)
            offsets.append(offset)
            offset += int(np.ceil(n / 8) * 8)
        offsets.append(offset)
        self.register_buffer("offsets", torch.from_numpy(np.array(offsets, dtype=np.int32)))
        self.n_params = offsets[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(offset, level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)

    def __repr__(self):
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"per_level_scale={self.per_level_scale:.4f} params={tuple(self.embeddings.shape)} gridtype={self.gridtype} "
                f"align_corners={self.align_corners}")

    def forward(self, inputs, bound=1):
        inputs = (inputs + bound) / (2 * bound)
        prefix = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        out = grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution, inputs.requires_grad,
                          self.gridtype_id, self.align_corners)
        return out.view(prefix + [self.output_dim])
