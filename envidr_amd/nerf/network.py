"""`NeRFNetwork`: ENVIDR's decomposed SDF + environment-lit shading model as a parameter container
with the reference's module / parameter names (`encoder.embeddings`, `sdf_net.N.weight`,
`sdf_density.beta`, `env_net`, `diffuse_net`, `color_net`, `renv_net`, buffers `density_bitfield`,
`aabb_infer`), so a reference checkpoint's `state_dict` loads with `load_state_dict(strict=False)`.

`forward_sigma` / `forward_color` restate nerf/network.py:381-698 for the configuration family the
shipped configs use (SDF + Laplace density, unit-norm features, sigmoid colours, IDE-fed env MLP,
optional renv branch); they run on the HIP encoders + torch GEMMs and exist for drop-in
compatibility and as the operator-level path.  The fast inference path does not call them: it hands
the same parameters to the fused persistent kernel (`_build_fused`).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import fused as _fused
from ..encoding import get_encoder
from .activation import trunc_exp
from .renderer import NeRFRenderer


class LaplaceDensity(nn.Module):
    """sigma = (1 / beta) * Laplace(0, beta).cdf(-sdf)   (reference network.py:26-44)"""

    def __init__(self, beta, beta_min=0.0001, beta_max=1.0):
        super().__init__()
        self.beta = nn.Parameter(torch.tensor(float(beta)))
        self.beta_min, self.beta_max = beta_min, beta_max

    def get_beta(self):
        clamped = torch.clamp(self.beta.detach(), self.beta_min, self.beta_max)
        return self.beta + (clamped - self.beta.detach())      # value = clamp(beta), gradient = d beta

    def density_func(self, sdf, beta=None, alpha=None):
        beta = self.get_beta() if beta is None else beta
        alpha = 1 / beta if alpha is None else alpha
        return alpha * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))

    def forward(self, sdf, beta=None):
        return self.density_func(sdf, beta=beta)


class NeuSDensity(nn.Module):
    """alpha of a sample from the sdf at its two section ends (NeuS; reference network.py:46-102): the value the compositors then take
    as `input_alpha` instead of a density"""

    def __init__(self, init_val, base_steps=1024, neus_n_detach=False):
        super().__init__()
        self.variance = nn.Parameter(torch.tensor(float(init_val)))
        self.base_steps, self.neus_n_detach = base_steps, neus_n_detach

    def get_variance(self):
        return self.variance

    def forward(self, sdf, dirs, dists, gradients, cos_anneal_ratio=1.0):
        inv_s = torch.exp(self.variance * 10.0).clip(1e-6, 1e6)
        if gradients is not None:
            gradients = gradients.detach() if self.neus_n_detach else gradients
            true_cos = (dirs * gradients).sum(-1, keepdim=True)
            iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio) + F.relu(-true_cos) * cos_anneal_ratio)
            next_sdf = sdf + iter_cos.squeeze() * dists * 0.5
            prev_sdf = sdf - iter_cos.squeeze() * dists * 0.5
        else:
            next_sdf, prev_sdf = sdf - dists * 0.5, sdf + dists * 0.5
        prev_cdf, next_cdf = torch.sigmoid(prev_sdf * inv_s), torch.sigmoid(next_sdf * inv_s)
        return ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).clip(0.0, 1.0)


def _mlp(dims, bias=True):
    return nn.ModuleList([nn.Linear(dims[i], dims[i + 1], bias=bias) for i in range(len(dims) - 1)])


ENV_MLP_OPERATOR_MIN_ROWS = 4096      # below this the four torch GEMMs are not slower than packing / launching the operator
WEIGHT_GRAD_OPERATOR_MIN_ROWS = 16384 # training: from this many rows the weight gradient of a layer goes to envidr_linear_weight_grad
# The shading networks of the training branch (environment, diffuse, colour heads) run as ONE autograd node each (_ReluMlp) from that many rows on.
# That node is once-differentiable: a loss that differentiates the colours with create_graph=True (a gradient penalty through the environment /
# diffuse / colour networks) raises there, where the reference's nn.Sequential works -- set SHADING_MLP_SINGLE_NODE = False for such a loss (the
# per-layer nodes below are twice differentiable).  Its ReLU (the ROWS_BIAS_RELU epilogue of envidr_linear_rows) propagates a NaN
# pre-activation like torch.relu; only the INFERENCE operator of the environment network (ds_max_f32 against 0) flushes NaN to 0.
SHADING_MLP_SINGLE_NODE = True


def _input_gradient_only():
    """inside hashencoder.input_gradient_only() (the normals pass: autograd.grad w.r.t. the positions, create_graph) a layer's backward
    skips the weight / bias gradient that autograd.grad would throw away -- a custom Function cannot see which of its gradients are wanted"""
    from ..hashencoder import hashgrid
    return bool(getattr(hashgrid._INPUT_GRADIENT_ONLY, "on", False))


def _rows_product(x, W, bias=None, relu=False, mask_act=None):
    """epilogue(x [M, K] @ W [N, K]^T) for a big batch of rows: envidr_linear_rows (csrc/linear_rows.hip, fp32 MFMA, W by its strides so a
    transposed view costs no copy, bias / ReLU / ReLU-gradient in the epilogue) wherever its operands qualify -- at 146 k rows it is 1.0 .. 3x
    the library GEMM plus its elementwise passes on every layer shape of the shipped networks (profiles/r05m/linear_rows_probe.txt); torch
    otherwise (a reduced width that is not a multiple of 4: the colour heads' 3 outputs on the way back)."""
    if _fused.linear_rows_supported(x, W):
        return _fused.linear_rows(x, W, bias=bias, relu=relu, mask_act=mask_act)
    y = x @ W.t() if bias is None else torch.addmm(bias, x, W.t())
    if relu:
        y = torch.relu_(y)
    if mask_act is not None:
        y = y * (mask_act > 0)
    return y


class _ReluMlp(torch.autograd.Function):
    """A whole ReLU MLP (x, W0, b0, W1, b1, ...) -> y as ONE autograd node, for the shading networks of the training branch (environment, diffuse,
    colour heads: differentiated once, by loss.backward() -- the twice-differentiated SDF network keeps the per-layer pair below).  Forward: one
    envidr_linear_rows per layer, bias and ReLU in its epilogue, the activations kept.  Backward: per layer one envidr_linear_weight_grad
    (dW and db in the same pass) and one envidr_linear_rows that carries the gradient to the layer below with the ReLU mask in its epilogue --
    instead of torch's addmm, clamp_min, threshold_backward, mm, mm, sum per layer (reference: plain nn.Sequential, network.py:524-698)."""

    @staticmethod
    def forward(ctx, x, *params):
        n = len(params) // 2
        acts, h = [x], x
        for i in range(n):
            h = _rows_product(h, params[2 * i], bias=params[2 * i + 1], relu=i != n - 1)
            acts.append(h)
        ctx.n = n
        ctx.save_for_backward(*params[0::2], *acts[:-1])
        return h

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        n = ctx.n
        Ws, acts = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        grads = [None] * (2 * n)
        g = gy.contiguous()
        gx = None
        for i in range(n - 1, -1, -1):
            if ctx.needs_input_grad[1 + 2 * i] or ctx.needs_input_grad[2 + 2 * i]:
                grads[2 * i], grads[2 * i + 1] = _fused.linear_weight_grad(acts[i], g, bias=True)
            if i > 0:
                g = _rows_product(g, Ws[i].t(), mask_act=acts[i])           # acts[i] = relu(layer i-1): its sign is the ReLU gradient
            elif ctx.needs_input_grad[0]:
                gx = _rows_product(g, Ws[0].t())
        return (gx, *grads)


class _RowsTimesMatrix(torch.autograd.Function):
    """x [M, K] , W [N, K]  ->  x W^T [M, N] for a batch of 10^4 .. 10^6 rows.  Forward and input gradient are library GEMMs (plenty of
    row parallelism); the gradient w.r.t. W -- an [N, K] result reduced over all rows, which a library GEMM tiles by its RESULT only
    (32 workgroups on 256 CUs: 0.4 ms per environment-MLP layer of a 144 k-sample batch) -- is _WeightGrad, i.e.
    envidr_linear_weight_grad.  The pair is closed under differentiation: each one's backward is made of the two, so the SDF network's
    twice-differentiated layers (normals with create_graph, then the loss) take the same path as the shading networks'."""

    @staticmethod
    def forward(ctx, x, W):
        ctx.save_for_backward(x, W)
        return _rows_product(x, W)

    @staticmethod
    def backward(ctx, gy):
        x, W = ctx.saved_tensors
        gx = _RowsTimesMatrix.apply(gy, W.t()) if ctx.needs_input_grad[0] else None
        gW = _WeightGrad.apply(x, gy) if ctx.needs_input_grad[1] and not _input_gradient_only() else None
        return gx, gW


class _WeightGrad(torch.autograd.Function):
    """x [M, K], gy [M, N]  ->  gy^T x [N, K]: envidr_linear_weight_grad (csrc/linear_grad.hip: the reduction over the rows split across
    the chip on the fp32 matrix cores, partial sums added in a fixed order)."""

    @staticmethod
    def forward(ctx, x, gy):
        ctx.save_for_backward(x, gy)
        return _fused.linear_weight_grad(x, gy, bias=False)[0]

    @staticmethod
    def backward(ctx, G):
        x, gy = ctx.saved_tensors
        gx = _RowsTimesMatrix.apply(gy, G.t()) if ctx.needs_input_grad[0] else None                    # gy G      [M, K]
        ggy = _RowsTimesMatrix.apply(x, G) if ctx.needs_input_grad[1] else None                        # x G^T    [M, N]
        return gx, ggy


class _Affine(torch.autograd.Function):
    """x W^T + b with the bias gradient taken from the same kernel pass as the weight gradient (envidr_linear_weight_grad sums the columns
    of gy on the way): no separate reduction over the batch per layer.  Differentiable to any order like the pair above."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        return _rows_product(x, W, bias=b)

    @staticmethod
    def backward(ctx, gy):
        x, W = ctx.saved_tensors
        gx = _RowsTimesMatrix.apply(gy, W.t()) if ctx.needs_input_grad[0] else None
        gW = gb = None
        if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and not _input_gradient_only():
            gW, gb = _WeightBiasGrad.apply(x, gy)
        return gx, gW, gb


class _WeightBiasGrad(torch.autograd.Function):
    """(gy^T x [N, K], column sums of gy [N]) in one envidr_linear_weight_grad call"""

    @staticmethod
    def forward(ctx, x, gy):
        ctx.save_for_backward(x, gy)
        return _fused.linear_weight_grad(x, gy, bias=True)

    @staticmethod
    def backward(ctx, G, g_b):
        x, gy = ctx.saved_tensors
        gx = _RowsTimesMatrix.apply(gy, G.t()) if ctx.needs_input_grad[0] else None
        ggy = None
        if ctx.needs_input_grad[1]:
            ggy = _RowsTimesMatrix.apply(x, G)
            if g_b is not None:
                ggy = ggy + g_b
        return gx, ggy


def _linear(lin, h, first_order_only=False):
    """nn.Linear; with autograd recording on a GPU batch of >= WEIGHT_GRAD_OPERATOR_MIN_ROWS rows (the training branch; the SDF network under
    the inference loop's autograd normals) as x W^T + b through envidr_linear_rows with the big-batch weight gradient -- which the normals
    pass, asking for the position gradient only, never forms (_input_gradient_only)"""
    if (torch.is_grad_enabled() and h.is_cuda and h.dtype == torch.float32 and lin.weight.dtype == torch.float32
            and lin.weight.requires_grad and h.numel() // max(h.shape[-1], 1) >= WEIGHT_GRAD_OPERATOR_MIN_ROWS):
        h2 = h.reshape(-1, h.shape[-1])
        y = _RowsTimesMatrix.apply(h2, lin.weight) if lin.bias is None else _Affine.apply(h2, lin.weight, lin.bias)
        return y.reshape(*h.shape[:-1], lin.weight.shape[0])
    if (not torch.is_grad_enabled() and h.is_cuda and h.dtype == torch.float32 and lin.weight.dtype == torch.float32
            and h.numel() // max(h.shape[-1], 1) >= ENV_MLP_OPERATOR_MIN_ROWS):
        # inference (the operator loop's shading heads): the same product, no graph
        return _rows_product(h.reshape(-1, h.shape[-1]), lin.weight, bias=lin.bias).reshape(*h.shape[:-1], lin.weight.shape[0])
    return lin(h)


def _run_mlp(net, h, first_order_only=False):
    # inference on the GPU, a shape the environment-MLP operator is built for (csrc/fused_render.hip k_env_mlp: the pass the fused
    # shading kernels run): one launch instead of four GEMMs and three ReLU passes over [M, H] activations.  With autograd
    # recording (training, or normals that need a graph) the torch layers below run, like the reference's.
    # One semantic difference, on non-finite activations only: the operator's ReLU is the LDS unit's ds_max_f32 against 0, which returns 0
    # for a NaN input (IEEE maxNum) where torch.relu propagates it -- a diverged environment MLP shows up as finite colours here, as NaN
    # in the torch chain.  ENV_MLP_OPERATOR_MIN_ROWS = a huge number keeps the torch chain.
    if (not torch.is_grad_enabled() and h.is_cuda and h.dtype == torch.float32 and h.numel() // max(h.shape[-1], 1) >= ENV_MLP_OPERATOR_MIN_ROWS
            and _fused.env_mlp_supported(net)):
        return _fused.env_mlp_forward(net, h)
    if (first_order_only and SHADING_MLP_SINGLE_NODE and torch.is_grad_enabled() and h.is_cuda and h.dtype == torch.float32 and h.numel() // max(h.shape[-1], 1) >= WEIGHT_GRAD_OPERATOR_MIN_ROWS
            and all(lin.training and lin.bias is not None and lin.weight.dtype == torch.float32 and lin.weight.requires_grad for lin in net)):
        params = [p for lin in net for p in (lin.weight, lin.bias)]
        return _ReluMlp.apply(h.reshape(-1, h.shape[-1]), *params).reshape(*h.shape[:-1], net[-1].weight.shape[0])
    if (not torch.is_grad_enabled() and h.is_cuda and h.dtype == torch.float32 and h.numel() // max(h.shape[-1], 1) >= ENV_MLP_OPERATOR_MIN_ROWS
            and all(lin.bias is not None and lin.weight.dtype == torch.float32 for lin in net)):
        # inference, a shape the environment-MLP operator is not built for: per layer one product with bias and ReLU in its epilogue
        h2 = h.reshape(-1, h.shape[-1])
        for i, lin in enumerate(net):
            h2 = _rows_product(h2, lin.weight, bias=lin.bias, relu=i != len(net) - 1)
        return h2.reshape(*h.shape[:-1], net[-1].weight.shape[0])
    for i, lin in enumerate(net):
        h = _linear(lin, h, first_order_only)
        if i != len(net) - 1:
            h = F.relu(h)
    return h


def _feat_act(x, kind):
    """geo_feat_act / env_feat_act (reference nerf/network.py:432-440, 538-546, 597-605, 646-654); any other name leaves the
    features as they are, as the reference's if / elif chains do"""
    if kind == "unitNorm":
        return F.normalize(x, dim=-1)
    if kind == "tanh":
        return torch.tanh(x)
    if kind == "instanceNorm":
        return (x - x.mean(dim=-1, keepdim=True)) / torch.sqrt(x.var(dim=-1, keepdim=True) + 1e-5)      # torch.var: unbiased
    return x


class NeRFNetwork(NeRFRenderer):
    def __init__(self, encoding="hashgrid", encoding_dir="sphere_harmonics", encoding_bg="hashgrid", num_layers=2, hidden_dim=64,
                 geo_feat_dim=15, num_layers_color=3, hidden_dim_color=64, num_layers_bg=2, hidden_dim_bg=64, bound=1,
                 num_levels=16, roughness_bias=-1, opt=None, env_opt=None, **kwargs):
        super().__init__(bound, opt=opt, env_opt=env_opt, **kwargs)
        # use_sdf = False (the plain-NeRF density branch, reference network.py:424-429,519): the geometry network's first output goes through
        # trunc_exp and IS the density; normals are the negated density gradient.  Operator path only: the fused kernels are built for the SDF
        # family (supports_fused asks for it).
        self.num_layers, self.hidden_dim, self.geo_feat_dim = num_layers, hidden_dim, geo_feat_dim
        self._encoding_dir = encoding_dir
        self.num_layers_color, self.hidden_dim_color = num_layers_color, hidden_dim_color
        self.roughness_bias = roughness_bias
        self.encoder, self.in_dim = get_encoder(opt.encoding_pos, level_dim=opt.level_dim,
                                                desired_resolution=bound * opt.desired_resolution,
                                                base_resolution=opt.base_resolution, num_levels=num_levels,
                                                log2_hashmap_size=opt.log2_hashmap_size, multires=opt.multires)
        # Laplace density (VolSDF style; every shipped config) or the NeuS section alpha (reference network.py:142-149); none without use_sdf
        if opt.use_sdf:
            self.sdf_density = (NeuSDensity(opt.init_variance, opt.max_steps, opt.neus_n_detach) if opt.use_neus_sdf
                                else LaplaceDensity(opt.init_beta, opt.beta_min, opt.beta_max))
        # material-conditioned SDF input (reference network.py:165-175): in the env-sphere mode the dataset's varying material parameters
        # -- roughness, metallic, base colour -- are concatenated to the hash features
        self.in_roughness = self.in_metallic = self.in_base_color = 0
        if opt.env_sph_mode:
            if env_opt is None:
                raise ValueError("env_sph_mode needs env_opt (vary_roughness / vary_metallic / vary_base_color, env_images_names)")
            self.in_roughness, self.in_metallic, self.in_base_color = \
                int(env_opt.vary_roughness), int(env_opt.vary_metallic), 3 * int(env_opt.vary_base_color)
        elif getattr(opt, "render_env_on_sphere", False):
            self.in_roughness, self.in_metallic, self.in_base_color = 1, 1, 3
        self.embed_dim = self.in_roughness + self.in_metallic + self.in_base_color
        self.w_material = self.embed_dim > 0
        out_dim = 1 + geo_feat_dim + (int(opt.use_roughness) + int(opt.learn_indir_blend) if opt.ensemble_mlp else 0)
        # skip_layers (reference network.py:178-194, 417-418): layer l in the list takes cat([h, x]) / sqrt(2), so the layer before it is
        # in_dim narrower; geometric_init (:153-159, 196-222): the IDR sphere initialisation -- biases forced on, weight-normalised layers
        # (state_dict keys weight_g / weight_v, like the reference's nn.utils.weight_norm), Softplus(beta = 100) instead of ReLU
        self.skip_layers = [int(l) for l in (opt.skip_layers or [])]
        self.geometric_init = bool(opt.geometric_init)
        first_in = self.in_dim + self.embed_dim
        layers = []
        for l in range(num_layers):
            d_in = first_in if l == 0 else hidden_dim
            d_out = out_dim if l == num_layers - 1 else (hidden_dim - first_in if l + 1 in self.skip_layers else hidden_dim)
            lin = nn.Linear(d_in, d_out, bias=self.geometric_init or opt.mlp_bias)
            if self.geometric_init:
                with torch.no_grad():
                    if l == num_layers - 1:
                        sign = -1.0 if opt.inside_outside else 1.0
                        nn.init.normal_(lin.weight, mean=sign * np.sqrt(np.pi) / np.sqrt(d_in), std=0.0001)
                        nn.init.constant_(lin.bias, -sign * opt.geo_init_bias)
                    elif first_in > 3 and l == 0:
                        nn.init.constant_(lin.bias, 0.0)
                        nn.init.constant_(lin.weight[:, 3:], 0.0)
                        nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(d_out))
                    else:
                        nn.init.constant_(lin.bias, 0.0)
                        nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(d_out))
                        if l in self.skip_layers:
                            nn.init.constant_(lin.weight[:, -(first_in - 3):], 0.0)
                lin = nn.utils.weight_norm(lin)
            layers.append(lin)
        self.sdf_net = nn.ModuleList(layers)
        self.sdf_act = nn.Softplus(beta=100) if self.geometric_init else nn.ReLU()
        if opt.use_roughness and not opt.ensemble_mlp:
            self.roughness_layer = nn.Linear(geo_feat_dim, 1)

        self.encoder_dir, self.in_dim_dir = get_encoder(encoding_dir, multires=opt.multires_dir, degree=opt.sh_degree)
        self.in_normal_dim = self.in_refdir_dim = 0
        if self.use_normal_with_mlp:
            self.encoder_normal, self.in_normal_dim = get_encoder(encoding_dir, multires=opt.multires_normal, degree=opt.sh_degree)
        if self.use_reflected_dir:
            self.encoder_refdir, self.in_refdir_dim = get_encoder(opt.encoding_ref, multires=opt.multires_refdir, degree=opt.sh_degree)
            self.diffuse_encoder_refdir, self.in_refdir_dim_diffuse = get_encoder(opt.encoding_ref, multires=opt.multires_refdir,
                                                                                  degree=opt.sh_degree_diffuse)
        self.use_viewdir = not opt.wo_viewdir
        if not self.use_viewdir:
            self.in_dim_dir = 0

        self.use_env_net = opt.use_env_net
        self.env_net = self.env_nets = self.renv_net = None
        if self.use_env_net:
            assert self.use_reflected_dir, "use_env_net requires use_reflected_dir"
            env_dims = [self.in_refdir_dim] + [opt.hidden_dim_env] * (opt.num_layers_env - 1) + [opt.env_feat_dim]
            if opt.env_sph_mode:
                # one environment MLP per environment of the dataset, selected by env_net_index (reference network.py:290-295, 530, 590)
                self.env_nets = nn.ModuleList([_mlp(env_dims, bias=not opt.env_wo_bias) for _ in env_opt.env_images_names])
            else:
                self.env_net = _mlp(env_dims, bias=not opt.env_wo_bias)
            if opt.split_diffuse_env and not opt.env_sph_mode:
                self.diffuse_env_net = _mlp([self.in_refdir_dim_diffuse] + [opt.hidden_dim_env_diffuse] * (opt.num_layers_env - 1)
                                            + [opt.env_feat_dim], bias=not opt.env_wo_bias)
            self.in_refdir_dim = opt.env_feat_dim
            if opt.use_renv:
                self.renv_net = _mlp([4, 64, 64, 64, opt.env_feat_dim], bias=not opt.env_wo_bias)
        if opt.use_diffuse:
            d_in = geo_feat_dim + (opt.env_feat_dim if opt.diffuse_with_env and opt.diffuse_env_fusion == "concat" else 0)
            self.diffuse_net = _mlp([d_in] + [opt.hidden_dim_diffuse] * (opt.num_layers_diffuse - 1) + [3], bias=True)
        self.n_dot_viewdir_dim = 1 if self.use_n_dot_viewdir else 0
        c_in = self.in_dim_dir + geo_feat_dim + self.in_normal_dim + self.in_refdir_dim + self.n_dot_viewdir_dim
        self.color_net = _mlp([c_in] + [hidden_dim_color] * (num_layers_color - 1) + [3], bias=opt.mlp_bias)
        gain = nn.init.calculate_gain("relu")
        for net in [None if self.geometric_init else self.sdf_net, self.env_net, *(self.env_nets or []), self.renv_net,
                    getattr(self, "diffuse_net", None), self.color_net]:
            if net is not None and opt.net_init == "xavier_uniform":
                for lin in net:
                    nn.init.xavier_uniform_(lin.weight, gain=gain)
                    if lin.bias is not None:
                        nn.init.zeros_(lin.bias)
        if opt.use_diffuse and opt.mlp_bias:
            self.color_net[-1].bias.data -= np.log(3)      # lower specular at initialisation
        # background model on the sphere of radius bg_radius (reference network.py:343-367): 2-D hash grid of the sphere
        # coordinates + SH(4) of the view direction -> bias-free MLP -> rgb
        self.bg_net = None
        if self.bg_radius > 0:
            self.num_layers_bg, self.hidden_dim_bg = num_layers_bg, hidden_dim_bg
            self.encoder_bg, self.in_dim_bg = get_encoder(encoding_bg, input_dim=2, num_levels=4, log2_hashmap_size=19, desired_resolution=2048)
            self.encoder_dir_bg, self.in_dim_dir_bg = get_encoder("sphere_harmonics", degree=4)
            self.bg_net = _mlp([self.in_dim_bg + self.in_dim_dir_bg] + [hidden_dim_bg] * (num_layers_bg - 1) + [3], bias=False)
        self.roughness = opt.default_roughness
        self.blend_weight = None
        self.metallic = 1.0
        self.c_diffuse = self.c_specular = 0

    @classmethod
    def from_scene(cls, scene, opt=None, device="cuda"):
        """a model holding the parameters of an `envidr_amd.scenes.SceneParams` (synthetic benchmark / test scenes),
        loaded through the state_dict keys a reference checkpoint uses"""
        from .options import toaster_options
        opt = opt or toaster_options()
        m = cls(encoding="hashgrid", encoding_dir=opt.encoding_dir, bound=opt.bound, cuda_ray=opt.cuda_ray, density_scale=1,
                min_near=opt.min_near, density_thresh=opt.density_thresh, bg_radius=opt.bg_radius, use_sdf=opt.use_sdf,
                hidden_dim=opt.hidden_dim, num_layers=opt.num_layers, num_layers_color=opt.num_layers_color,
                hidden_dim_color=opt.hidden_dim_color, num_levels=opt.num_levels, geo_feat_dim=opt.geo_feat_dim, opt=opt)
        sd = {"encoder.embeddings": torch.from_numpy(scene.table), "sdf_density.beta": torch.tensor(scene.beta),
              "density_bitfield": torch.from_numpy(scene.bitfield)}
        for name, attr in [("sdf", "sdf_net"), ("env", "env_net"), ("diffuse", "diffuse_net"), ("specular", "color_net"),
                           ("renv", "renv_net")]:
            for i, (W, b) in enumerate(scene.mlps.get(name, [])):
                sd[f"{attr}.{i}.weight"] = torch.from_numpy(W)
                sd[f"{attr}.{i}.bias"] = torch.from_numpy(b)
        res = m.load_state_dict(sd, strict=False)
        if res.unexpected_keys:
            raise ValueError(f"scene parameters the model has no slot for: {res.unexpected_keys}")
        return m.to(device).eval()

    # ---- geometry ---------------------------------------------------------------------------------
    def material_vector(self, material) -> list:
        """the material parameters in the order the SDF network's extra inputs take them (reference network.py:369-379):
        [roughness][metallic][r, g, b]"""
        v = []
        if self.in_roughness:
            v.append(float(material["roughness"]))
        if self.in_metallic:
            v.append(float(material["metallic"]))
        if self.in_base_color:
            v.extend(float(c) for c in list(material["color"])[:3])
        return v

    def concate_material_params(self, x, material):
        if material is None:
            raise ValueError("this model's SDF network takes material parameters: pass material={'roughness', 'metallic', 'color'}")
        m = torch.tensor(self.material_vector(material), dtype=x.dtype, device=x.device)
        return torch.cat([x, m.expand(*x.shape[:-1], m.shape[0])], dim=-1)

    def forward_geometry(self, xyz, material=None):
        from ..hashencoder import HashEncoder
        # Only when the caller says that nothing but the input gradient (normals) will be taken -- run_cuda's inference loop,
        # which detaches everything it gets back, sets `_normals_only` around its call -- is the table kept out of the graph
        # (autograd would otherwise zero-fill and scatter a 48.8 MB table gradient nobody reads, every loop iteration).  Any other
        # caller, in train() or eval() mode, gets the reference's graph: embeddings included.
        kw = {"table_grad": False} if (getattr(self, "_normals_only", False) and isinstance(self.encoder, HashEncoder)) else {}
        x = self.encoder(xyz, bound=self.bound, **kw)
        if self.opt.enabled_levels > 0:
            mask = torch.zeros(self.opt.num_levels, self.opt.level_dim, device=x.device)
            mask[: self.opt.enabled_levels] += 1
            x = x * mask.reshape(-1)
        if self.w_material:
            x = self.concate_material_params(x, material)
        if self.skip_layers or self.geometric_init:
            h = x
            for l, lin in enumerate(self.sdf_net):
                if l in self.skip_layers:
                    h = torch.cat([h, x], dim=-1) / np.sqrt(2)
                h = lin(h)
                if l != self.num_layers - 1:
                    h = self.sdf_act(h)
        else:
            h = _run_mlp(self.sdf_net, x)
        sdf, sigma = (h[..., 0], None) if self.use_sdf else (None, trunc_exp(h[..., 0]))
        g = self.geo_feat_dim
        geo_feat = _feat_act(h[..., 1:1 + g], self.opt.geo_feat_act)
        if self.opt.use_roughness and not self.opt.diffuse_only and not self.opt.bypass_roughness:
            raw = h[..., 1 + g:2 + g] if self.opt.ensemble_mlp else self.roughness_layer(geo_feat)
            if self.opt.learn_indir_blend and self.opt.ensemble_mlp:
                self.blend_weight = torch.sigmoid(h[..., 2 + g:3 + g])
            self.roughness = self.opt.roughness_act_scale * F.softplus(raw + self.roughness_bias) * self.opt.roughness_scale
        else:
            self.roughness = self.opt.default_roughness
        self.metallic = 1.0
        return sdf, sigma, geo_feat

    def forward_sigma(self, xyzs, material=None, **kwargs):
        sdfs, sigmas, geo_feats = self.forward_geometry(xyzs, material)
        normals = eikonal = None
        if not self.use_sdf:                                # reference network.py:519-520: the density is there already; its gradient gives the normals
            if kwargs.get("use_sdf_sigma_grad", False):
                normals, eikonal = self.compute_normal(sigmas, xyzs, self.opt.eikonal_loss)
            return sdfs, sigmas, geo_feats, normals, eikonal
        if kwargs.get("use_sdf_sigma_grad", False):
            normals, eikonal = self.compute_normal(sdfs, xyzs, self.opt.eikonal_loss)
        if self.opt.use_neus_sdf:                           # reference network.py:512-515: the "density" is the section alpha
            dists = kwargs.get("dists", None)
            dists = 2 * 3 ** 0.5 / self.sdf_density.base_steps if dists is None else dists
            sigmas = self.sdf_density(sdfs, dirs=kwargs.get("dirs", None), dists=dists, gradients=normals,
                                      cos_anneal_ratio=getattr(self.opt, "cos_anneal_ratio", 1.0))
        else:
            sigmas = self.sdf_density(sdfs)
        return sdfs, sigmas, geo_feats, normals, eikonal

    def density(self, x, **kwargs):
        sdf, sigma, geo_feat, normal, eik = self.forward_sigma(x, **kwargs)
        return {"sdf": sdf, "sigma": sigma, "geo_feat": geo_feat, "normal": normal, "sdf_gradients": eik}

    # ---- shading ----------------------------------------------------------------------------------
    def _env(self, net, enc):
        return _feat_act(_run_mlp(net, enc, first_order_only=True), self.opt.env_feat_act)

    def forward_color(self, geo_feat, d, normal=None, w_r=None, n_dot_w_o=None, use_specular_color=False, env_net_index=0,
                      n_env_enc=None, r_images=None, roughness=None):
        opt = self.opt
        # The diffuse side (encoded normal) and the specular side (encoded reflection) query the SAME environment network unless
        # split_diffuse_env: one evaluation over both batches -- half the launches (forward and backward), and a big batch's last, partly
        # filled round of workgroups once instead of twice (reference network.py:533-536 and 592-595 call it twice)
        e_normal = e_reflect = None
        if (opt.use_diffuse and opt.diffuse_with_env and not opt.diffuse_only and not opt.split_diffuse_env and self.use_env_net
                and n_env_enc is not None and w_r is not None and not opt.train_renv and n_env_enc.shape == w_r.shape):
            net = self.env_nets[env_net_index] if opt.env_sph_mode else self.env_net
            both = self._env(net, torch.cat([n_env_enc, w_r], 0))
            e_normal, e_reflect = both[:n_env_enc.shape[0]], both[n_env_enc.shape[0]:]
        if opt.use_diffuse:
            h = geo_feat
            if opt.diffuse_with_env:
                env_net = self.env_nets[env_net_index] if opt.env_sph_mode else (self.diffuse_env_net if opt.split_diffuse_env else self.env_net)
                e = e_normal if e_normal is not None else self._env(env_net, n_env_enc)
                h = {"concat": lambda: torch.cat([h, e], -1), "add": lambda: h + e, "mul": lambda: h * e}[opt.diffuse_env_fusion]()
            self.c_diffuse = torch.sigmoid(_run_mlp(self.diffuse_net, h, first_order_only=True)) * self.metallic
        else:
            self.c_diffuse = 0
        if opt.diffuse_only:
            self.c_specular = 0
            return (self.c_diffuse + self.c_specular) * opt.intensity_scale

        h = torch.cat([self.encoder_dir(d), geo_feat], -1) if self.use_viewdir else geo_feat
        if self.use_normal_with_mlp:
            h = torch.cat([h, normal], -1)
        branches, renv_mask, blend = {}, None, 1
        if w_r is not None and not opt.train_renv:
            env_net = self.env_nets[env_net_index] if (opt.env_sph_mode and self.use_env_net) else self.env_net
            branches["env"] = torch.cat([h, (e_reflect if e_reflect is not None else self._env(env_net, w_r)) if self.use_env_net else w_r], -1)
        if r_images is not None and opt.use_renv:
            renv_mask = roughness.squeeze() < opt.indir_roughness_thresh
            if r_images.shape[-1] == 4:
                vis = r_images[..., -1]
                r_images = r_images[..., :3] * vis[..., None].detach()
                renv_mask = renv_mask & (vis > 0.9)
            rough = roughness[renv_mask] / opt.roughness_scale
            remap = torch.sqrt(rough / 0.75)
            blend = 0.98 * self.blend_weight[renv_mask] if opt.learn_indir_blend else 0.95 * torch.sigmoid(80 * (remap - 0.18))
            e = _feat_act(_run_mlp(self.renv_net, torch.cat([r_images[renv_mask], remap], -1), first_order_only=True), opt.env_feat_act)
            branches["renv"] = torch.cat([h[renv_mask], e], -1)
        if not branches:
            branches["env"] = h
        colors = {}
        for k, hc in branches.items():
            if n_dot_w_o is not None:
                hc = torch.cat([hc, n_dot_w_o[renv_mask] if k == "renv" else n_dot_w_o], -1)
            colors[k] = torch.sigmoid(_run_mlp(self.color_net, hc, first_order_only=True))
        self.c_specular = colors["env"]
        if "renv" in colors:
            if opt.indir_only:
                self.c_specular = self.c_specular * 0
            mixed = self.c_specular[renv_mask] * blend + colors["renv"] * (1 - blend)
            self.c_specular = self.c_specular.masked_scatter(renv_mask[:, None], mixed)
        return (self.c_diffuse + self.c_specular) * opt.intensity_scale

    def forward(self, x, d, normal=None, w_r=None, n_dot_w_o=None):
        sdf, sigma, geo_feat, _, _ = self.forward_sigma(x)
        return sdf, sigma, self.forward_color(geo_feat, d, normal, w_r, n_dot_w_o)

    def background(self, x, d):
        """rgb of the background sphere: x [N,2] sphere coordinates in [-1,1] (raymarching.sph_from_ray), d [N,3] view
        directions (reference network.py:727-742)"""
        h = torch.cat([self.encoder_dir_bg(d), self.encoder_bg(x)], dim=-1)
        return torch.sigmoid(_run_mlp(self.bg_net, h)) if self.opt.color_act == "sigmoid" else _run_mlp(self.bg_net, h)

    def color(self, x, d, mask=None, geo_feat=None, normal=None, w_r=None, n_dot_w_o=None, **kwargs):
        """rgb of the samples selected by `mask` (all when None), zeros elsewhere (reference network.py:745-770)"""
        if mask is None:
            return self.forward_color(geo_feat, d, normal, w_r, n_dot_w_o)
        rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
        if not mask.any():
            return rgbs
        pick = lambda t: None if t is None else t[mask]
        rgbs[mask] = self.forward_color(geo_feat[mask], d[mask], pick(normal), pick(w_r), pick(n_dot_w_o)).to(rgbs.dtype)
        return rgbs

    # ---- fused path -------------------------------------------------------------------------------
    def supports_fused(self, r_images=None, geometry_only=False, **kwargs) -> bool:
        """configurations the fused persistent kernel implements (everything else uses the operator loop)"""
        o = self.opt
        hash_ok = o.encoding_pos == "hashgrid_diff" and o.level_dim == 2 and o.num_levels == 16      # the fused kernels are built for 16 levels
        net_ok = (o.num_layers == 3 and o.hidden_dim == 64 and o.geo_feat_dim == 12 and o.ensemble_mlp and o.use_roughness
                  and o.learn_indir_blend and o.mlp_bias and o.geo_feat_act == "unitNorm" and o.env_feat_act == "unitNorm"
                  and self.use_sdf and not o.use_neus_sdf and not self.geometric_init and not self.skip_layers and not self.w_material)
        shade_ok = (o.use_diffuse and not o.diffuse_only and o.diffuse_with_env and o.diffuse_env_fusion == "concat"
                    and not o.split_diffuse_env and o.use_env_net and not o.env_wo_bias and o.num_layers_env == 4
                    and o.env_feat_dim == 12 and (o.sh_degree, o.hidden_dim_env) in [(5, 256), (4, 160), (5, 128), (4, 128)]
                    and o.wo_viewdir and o.normal_with_mlp and o.multires_normal == 0 and o.use_n_dot_viewdir
                    and o.use_reflected_dir and o.encoding_ref == "integrated_dir" and o.num_layers_diffuse == 2
                    and o.hidden_dim_diffuse == 32 and o.num_layers_color == 3 and o.hidden_dim_color == 64
                    and o.color_act == "sigmoid" and o.normal_anneal_ratio >= 1)
        # the no-environment family (BASELINE configs[1]): SH-encoded view direction and normal into the specular head
        plain_ok = (o.use_diffuse and not o.diffuse_only and not o.diffuse_with_env and not o.use_env_net and not o.use_reflected_dir
                    and not o.wo_viewdir and self._encoding_dir == "sphere_harmonics" and o.sh_degree == 4 and o.normal_with_mlp
                    and o.use_n_dot_viewdir and o.num_layers_diffuse == 2 and o.hidden_dim_diffuse == 32 and o.num_layers_color == 3
                    and o.hidden_dim_color == 64 and o.color_act == "sigmoid" and o.normal_anneal_ratio >= 1)
        # reflected-radiance branch of the main indirect pass: renv MLP 4-64-64-64-12 with the learnt blend
        renv_ok = r_images is None or (shade_ok and o.use_renv and self.renv_net is not None and o.learn_indir_blend
                                       and not o.indir_only and not o.train_renv and r_images.shape[-1] == 4)
        return bool(hash_ok and net_ok and (shade_ok or plain_ok) and renv_ok and not self.training)

    # ---- env-sphere mode on the fused kernels --------------------------------------------------------
    def supports_fused_sph(self) -> bool:
        """run_sph as shell samples -> envidr_geometry_eval -> envidr_shade_samples -> envidr_composite_shell: the hash grid and the
        network shapes the fused kernels are built for (configs/neural_renderer.ini is one of them)"""
        o = self.opt
        hash_ok = o.encoding_pos == "hashgrid_diff" and o.level_dim == 2 and o.num_levels == 16
        net_ok = (o.num_layers == 3 and o.hidden_dim == 64 and o.geo_feat_dim == 12 and o.ensemble_mlp and o.use_roughness and o.mlp_bias
                  and o.geo_feat_act == "unitNorm" and o.env_feat_act == "unitNorm" and o.enabled_levels <= 0 and not o.bypass_roughness
                  and o.normal_anneal_ratio >= 1 and self.use_sdf and not o.use_neus_sdf and not self.geometric_init and not self.skip_layers)
        shade_ok = (o.use_diffuse and not o.diffuse_only and o.diffuse_with_env and o.diffuse_env_fusion == "concat"
                    and not o.split_diffuse_env and o.use_env_net and self.env_nets is not None and not o.env_wo_bias and o.num_layers_env == 4
                    and o.env_feat_dim == 12 and (o.sh_degree, o.hidden_dim_env) in [(5, 256), (4, 160), (5, 128), (4, 128)]
                    and o.wo_viewdir and o.normal_with_mlp and o.multires_normal == 0 and o.use_n_dot_viewdir
                    and o.use_reflected_dir and o.encoding_ref == "integrated_dir" and o.num_layers_diffuse == 2
                    and o.hidden_dim_diffuse == 32 and o.num_layers_color == 3 and o.hidden_dim_color == 64 and o.color_act == "sigmoid")
        return bool(hash_ok and net_ok and shade_ok)

    def _sdf_layers_for_material(self, material):
        """the SDF network as the geometry kernel takes it (2L -> 64 -> 64 -> 15): the material parameters are constants of a render
        call, so their columns of the first layer fold into its bias -- W1 [x | m] + b1 = W1[:, :2L] x + (b1 + W1[:, 2L:] m) -- and the
        input gradient (the normal) does not see them; the last layer is padded with zero rows up to the kernel's 15 outputs"""
        W1, b1 = self.sdf_net[0].weight.detach().double(), self.sdf_net[0].bias.detach().double()
        feat = self.in_dim
        if self.w_material:
            m = torch.tensor(self.material_vector(material), dtype=torch.float64, device=W1.device)
            b1 = b1 + W1[:, feat:] @ m
        W3, b3 = self.sdf_net[2].weight.detach(), self.sdf_net[2].bias.detach()
        if W3.shape[0] < 15:
            W3 = torch.cat([W3, W3.new_zeros(15 - W3.shape[0], W3.shape[1])])
            b3 = torch.cat([b3, b3.new_zeros(15 - b3.shape[0])])
        return [(W1[:, :feat].float().contiguous(), b1.float()), (self.sdf_net[1].weight.detach(), self.sdf_net[1].bias.detach()), (W3, b3)]

    def fused_sph_renderer(self, env_net_index: int, material):
        """the fused renderer of the env-sphere mode for one environment MLP, its SDF weights set for `material`"""
        from ..fused import FusedRenderer
        key = tuple(self.material_vector(material)) if self.w_material else ()
        # The renderer copies the packed MLPs and beta at construction (the hash table is used in place).  cuda_ray is off in this mode,
        # so update_extra_state() -- the only automatic invalidate_fused() -- never runs: the weights' own (address, version) pairs are
        # part of the cache key, as in fused.env_mlp_forward, so an optimizer step between two evaluations rebuilds the renderer.
        nets = (self.sdf_net, self.env_nets[env_net_index], self.diffuse_net, self.color_net)
        stamp = tuple((p.data_ptr(), p._version) for net in nets for p in net.parameters()) + ((self.sdf_density.beta.data_ptr(), self.sdf_density.beta._version),
                                                                                                 self.encoder.embeddings.data_ptr())
        entry = self._fused_sph.get(env_net_index)
        if entry is not None and entry[2] != stamp:
            entry = None
        if entry is None:
            pairs = lambda net: [(l.weight.detach(), l.bias.detach()) for l in net]
            mlps = {"sdf": self._sdf_layers_for_material(material), "env": pairs(self.env_nets[env_net_index]),
                    "diffuse": pairs(self.diffuse_net), "specular": pairs(self.color_net)}
            dev = self.encoder.embeddings.device
            # (no occupancy grid in this mode: hits are analytic; the renderer only wants a bitfield of the right size)
            bitfield = torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8, device=dev)
            fr = FusedRenderer(bitfield, self.encoder.embeddings.detach(), self.encoder.offsets.cpu().numpy(), self.encoder.per_level_scale,
                               mlps, float(self.sdf_density.beta.detach()), self._fused_options(), device=dev)
            entry = self._fused_sph[env_net_index] = [fr, key, stamp]
        elif entry[1] != key:
            entry[0].update_sdf(self._sdf_layers_for_material(material))
            entry[1] = key
        return entry[0]

    def _fused_options(self):
        from ..fused import FusedOptions
        o = self.opt
        return FusedOptions(bound=self.bound, grid_size=self.grid_size, min_near=self.min_near, max_steps=o.max_steps,
                            dt_gamma=o.dt_gamma, T_thresh=o.T_thresh, density_scale=self.density_scale,
                            base_resolution=o.base_resolution, enabled_levels=o.enabled_levels, beta_min=o.beta_min,
                            beta_max=o.beta_max, roughness_bias=self.roughness_bias, roughness_act_scale=o.roughness_act_scale,
                            roughness_scale=o.roughness_scale, ide_degree=o.sh_degree, diffuse_kappa_inv=o.diffuse_kappa_inv,
                            light_intensity_scale=o.light_intensity_scale, intensity_scale=o.intensity_scale,
                            dir_sh_degree=0 if self.use_env_net else o.sh_degree, indir_roughness_thresh=o.indir_roughness_thresh)

    def _build_fused(self):
        from ..fused import FusedOptions, FusedRenderer
        o = self.opt
        fo = FusedOptions(bound=self.bound, grid_size=self.grid_size, min_near=self.min_near, max_steps=o.max_steps,
                          dt_gamma=o.dt_gamma, T_thresh=o.T_thresh, density_scale=self.density_scale,
                          base_resolution=o.base_resolution, enabled_levels=o.enabled_levels, beta_min=o.beta_min,
                          beta_max=o.beta_max, roughness_bias=self.roughness_bias, roughness_act_scale=o.roughness_act_scale,
                          roughness_scale=o.roughness_scale, ide_degree=o.sh_degree, diffuse_kappa_inv=o.diffuse_kappa_inv,
                          light_intensity_scale=o.light_intensity_scale, intensity_scale=o.intensity_scale,
                          dir_sh_degree=0 if self.use_env_net else o.sh_degree, indir_roughness_thresh=o.indir_roughness_thresh)
        pairs = lambda net: [(l.weight.detach(), l.bias.detach()) for l in net]
        mlps = {"sdf": pairs(self.sdf_net), "diffuse": pairs(self.diffuse_net), "specular": pairs(self.color_net)}
        if self.use_env_net:
            mlps["env"] = pairs(self.env_net)
            if self.renv_net is not None:
                mlps["renv"] = pairs(self.renv_net)
        return FusedRenderer(self.density_bitfield, self.encoder.embeddings.detach(), self.encoder.offsets.cpu().numpy(),
                             self.encoder.per_level_scale, mlps, float(self.sdf_density.beta.detach()), fo,
                             device=self.density_bitfield.device)
