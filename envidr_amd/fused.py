"""Host side of the fused inference renderer (include/envidr_render.h).

`FusedRenderer` owns the device-resident model state of one scene -- occupancy bitfield, hash
table, and the MLP weights re-packed once into the MFMA tile layout -- and renders ray batches with
a single persistent-kernel launch per call.  It is what `NeRFRenderer.render()` dispatches to for
the inference branch of `run_cuda` (nerf/render_func/cuda_ray.py:238-359); the Python here only
moves pointers.  No CPU fallback: without libenvidr_amd.so every call raises.
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib

MAX_LEVELS = 16
_FP = ctypes.c_void_p


class GeometryExport(ctypes.Structure):
    """mirror of `envidr_geometry_export`"""
    _fields_ = [("counter", _FP), ("capacity", ctypes.c_uint32), ("ray", _FP), ("idx", _FP), ("w", _FP), ("normal", _FP),
                ("geo_feat", _FP), ("roughness", _FP), ("slot", _FP), ("blend", _FP), ("image_width", ctypes.c_uint32),
                ("shade_list", _FP)]


class SamplesOut(ctypes.Structure):
    """mirror of `envidr_geometry_samples_out`"""
    _fields_ = [(n, _FP) for n in ("alpha", "sigma", "normal", "geo_feat", "roughness", "blend")]


class RenderDesc(ctypes.Structure):
    """mirror of `envidr_render_desc` (include/envidr_render.h) -- keep field order in sync"""
    _fields_ = [
        ("density_bitfield", _FP), ("bound", ctypes.c_float), ("cascades", ctypes.c_uint32), ("grid_size", ctypes.c_uint32),
        ("min_near", ctypes.c_float), ("max_steps", ctypes.c_uint32), ("dt_gamma", ctypes.c_float), ("T_thresh", ctypes.c_float),
        ("density_scale", ctypes.c_float), ("bg_color", ctypes.c_float),
        ("hash_table", _FP), ("hash_offsets", ctypes.c_int32 * (MAX_LEVELS + 1)), ("num_levels", ctypes.c_uint32),
        ("base_resolution", ctypes.c_uint32), ("log2_per_level_scale", ctypes.c_float), ("enabled_levels", ctypes.c_int32),
        ("sdf_blob", _FP), ("env_blob", _FP), ("head_blob", _FP), ("sdf_w3_row0", _FP),
        ("beta", ctypes.c_float), ("roughness_bias", ctypes.c_float), ("roughness_act_scale", ctypes.c_float),
        ("roughness_scale", ctypes.c_float),
        ("ide_degree", ctypes.c_uint32), ("env_hidden", ctypes.c_uint32),
        ("diffuse_kappa_inv", ctypes.c_float), ("light_intensity_scale", ctypes.c_float), ("intensity_scale", ctypes.c_float),
        ("has_env_rot", ctypes.c_int32), ("env_rot", ctypes.c_float * 9), ("dir_sh_degree", ctypes.c_uint32),
        ("geometry_only", ctypes.c_int32), ("r_images", _FP), ("renv_blob", _FP), ("spec2_blob", _FP),
        ("indir_roughness_thresh", ctypes.c_float), ("geometry_export", ctypes.POINTER(GeometryExport)),
        ("ray_cost", _FP), ("scratch", _FP), ("scratch_bytes", ctypes.c_uint64),
        ("has_aabb", ctypes.c_int32), ("aabb", ctypes.c_float * 6), ("ray_mask", _FP),
        ("env_split_blob", _FP), ("env_split_bias", _FP), ("env_features", _FP),
        ("sdf_geo_blob", _FP), ("image_width", ctypes.c_uint32), ("env_split_form", ctypes.c_uint32),
    ]


class RenderOut(ctypes.Structure):
    """mirror of `envidr_render_out`"""
    _fields_ = [("image", _FP), ("depth", _FP), ("weights_sum", _FP), ("normal_image", _FP), ("diffuse_image", _FP),
                ("specular_image", _FP), ("roughness_image", _FP), ("stats", _FP)]


@dataclass
class FusedOptions:
    """render knobs (defaults: configs/scenes/toaster.ini + nerf/options.py)"""
    bound: float = 1.0
    grid_size: int = 128
    min_near: float = 0.2
    max_steps: int = 1024
    dt_gamma: float = 0.0
    T_thresh: float = 1e-4
    density_scale: float = 1.0
    bg_color: float = 1.0
    base_resolution: int = 16
    enabled_levels: int = -1
    indir_roughness_thresh: float = 0.1
    dir_sh_degree: int = 0       # > 0: no environment MLP, SH-encoded view direction / normal (BASELINE configs[1])
    beta_min: float = 0.0005
    beta_max: float = 1.0
    roughness_bias: float = -1.0
    roughness_act_scale: float = 0.2
    roughness_scale: float = 1.0
    ide_degree: int = 5
    diffuse_kappa_inv: float = 0.64
    light_intensity_scale: float = 1.0
    intensity_scale: float = 1.0
    # arithmetic of the environment MLP: "fp32" (default, what every headline number uses) or "f16x2" -- fp16 matrix cores
    # with every operand carried as a (hi, lo) fp16 pair, fp32 accumulation (csrc/mlp_split2.hip.h; "f16x2_v1": the round-3 kernel of
    # csrc/mlp_split.hip.h, same bits); the heads stay fp32
    env_precision: str = "fp32"
    # records whose compositing weight alpha * T is exactly 0 in fp32 are not shaded (they contribute w * c = 0 whatever c is; the
    # reference shades them all): nothing on the synthetic benchmark scene, most of the samples of a trained scene (beta ~ 1e-3)
    skip_zero_weight: bool = True
    # per-sample geometry kernel: "32" = k_geo_eval32 (32 samples per wave, two waves per SIMD: the default, and the faster),
    # "16" = k_geo_eval16 (16-column MFMAs, three waves per SIMD; kept as a measured alternative, csrc/geo_eval16.hip.h)
    geometry_kernel: str = "32"


def _bind_render(lib):
    if getattr(lib, "_envidr_render_bound", False):
        return
    lib.envidr_packed_weight_floats.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32]
    lib.envidr_packed_weight_floats.restype = ctypes.c_uint32
    lib.envidr_packed_rowvec_floats.argtypes = [ctypes.c_uint32]
    lib.envidr_packed_rowvec_floats.restype = ctypes.c_uint32
    lib.envidr_pack_linear.argtypes = [_FP, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, _FP]
    lib.envidr_pack_linear.restype = ctypes.c_int
    lib.envidr_packed_layer_floats.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int]
    lib.envidr_packed_layer_floats.restype = ctypes.c_uint32
    lib.envidr_pack_layer.argtypes = [_FP, _FP, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, _FP]
    lib.envidr_pack_layer.restype = ctypes.c_int
    lib.envidr_sdf_geometry_floats.restype = ctypes.c_uint32
    lib.envidr_pack_sdf_geometry.argtypes = [_FP] * 7
    lib.envidr_pack_sdf_geometry.restype = ctypes.c_int
    lib.envidr_split_layer_halves.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32]
    lib.envidr_split_layer_halves.restype = ctypes.c_uint32
    lib.envidr_split_chunk_bytes.restype = ctypes.c_uint32
    lib.envidr_split_group.restype = ctypes.c_uint32
    lib.envidr_pack_layer_split.argtypes = [_FP, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, _FP]
    lib.envidr_pack_layer_split.restype = ctypes.c_int
    lib.envidr_env_split2_halves.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    lib.envidr_env_split2_halves.restype = ctypes.c_uint32
    lib.envidr_pack_env_split2.argtypes = [_FP] * 4 + [ctypes.c_uint32, ctypes.c_uint32, _FP]
    lib.envidr_pack_env_split2.restype = ctypes.c_int
    lib.envidr_pack_rowvec.argtypes = [_FP, ctypes.c_uint32, _FP]
    lib.envidr_pack_rowvec.restype = ctypes.c_int
    lib.envidr_render_rays.argtypes = [ctypes.POINTER(RenderDesc), _FP, _FP, ctypes.c_uint32, ctypes.POINTER(RenderOut), _FP, _FP]
    lib.envidr_render_rays.restype = ctypes.c_int
    lib.envidr_render_scratch_bytes.argtypes = [ctypes.c_uint32]
    lib.envidr_render_scratch_bytes.restype = ctypes.c_uint64
    lib.envidr_shade_samples.argtypes = [ctypes.POINTER(RenderDesc), _FP, _FP, _FP, ctypes.c_uint32, _FP, ctypes.c_uint32,
                                         ctypes.c_uint32, _FP, _FP, _FP]
    lib.envidr_shade_samples.restype = ctypes.c_int
    lib.envidr_composite_shaded.argtypes = [_FP, _FP, _FP, _FP, _FP, ctypes.c_uint32, ctypes.c_float, ctypes.c_float, _FP, _FP, _FP, _FP]
    lib.envidr_composite_shaded.restype = ctypes.c_int
    lib.envidr_shade_records.argtypes = [ctypes.POINTER(RenderDesc), ctypes.POINTER(GeometryExport), _FP, _FP, _FP, _FP]
    lib.envidr_shade_records.restype = ctypes.c_int
    lib.envidr_composite_records.argtypes = [ctypes.POINTER(GeometryExport), _FP, _FP, _FP, _FP, _FP, ctypes.c_uint32, ctypes.c_float,
                                             ctypes.c_float, _FP, _FP, _FP, _FP]
    lib.envidr_composite_records.restype = ctypes.c_int
    lib.envidr_geometry_eval.argtypes = [ctypes.POINTER(RenderDesc), _FP, _FP, ctypes.c_uint32, _FP, ctypes.POINTER(SamplesOut), _FP]
    lib.envidr_geometry_eval.restype = ctypes.c_int
    lib.envidr_geometry_probe.argtypes = [ctypes.POINTER(RenderDesc), _FP, ctypes.c_uint32, _FP, _FP, _FP, _FP, _FP]
    lib.envidr_geometry_probe.restype = ctypes.c_int
    lib.envidr_geometry_workspace_bytes.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    lib.envidr_geometry_workspace_bytes.restype = ctypes.c_uint64
    lib.envidr_geometry_pass.argtypes = [ctypes.POINTER(RenderDesc), _FP, _FP, ctypes.c_uint32, ctypes.POINTER(RenderOut),
                                         ctypes.POINTER(GeometryExport), _FP, ctypes.c_uint64, ctypes.c_uint32, _FP]
    lib.envidr_geometry_pass.restype = ctypes.c_int
    lib.envidr_env_mlp_forward.argtypes = [_FP, ctypes.c_uint32, ctypes.c_uint32, _FP, ctypes.c_uint32, _FP, _FP]
    lib.envidr_env_mlp_forward.restype = ctypes.c_int
    lib.envidr_linear_weight_grad_workspace_bytes.argtypes = [ctypes.c_uint32] * 3
    lib.envidr_linear_weight_grad_workspace_bytes.restype = ctypes.c_uint64
    lib.envidr_linear_weight_grad.argtypes = [_FP, _FP, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, _FP, _FP, ctypes.c_int, _FP, ctypes.c_uint64, _FP]
    lib.envidr_linear_weight_grad.restype = ctypes.c_int
    lib.envidr_linear_rows.argtypes = [_FP, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, _FP, ctypes.c_int64, ctypes.c_int64, ctypes.c_uint32, _FP, _FP,
                                       ctypes.c_uint32, ctypes.c_int, _FP, ctypes.c_uint32, _FP]
    lib.envidr_linear_rows.restype = ctypes.c_int
    lib.envidr_sphere_intersections.argtypes = [_FP, _FP, ctypes.c_uint32, ctypes.c_float, _FP, _FP, _FP, _FP]
    lib.envidr_sphere_intersections.restype = ctypes.c_int
    lib.envidr_shell_samples.argtypes = [_FP, _FP, _FP, _FP, _FP, _FP, ctypes.c_float, ctypes.c_uint32, ctypes.c_uint32, _FP, _FP, _FP, _FP]
    lib.envidr_shell_samples.restype = ctypes.c_int
    lib.envidr_composite_shell.argtypes = [_FP] * 10 + [ctypes.c_uint32] * 3 + [ctypes.c_float] * 2 + [_FP] * 8
    lib.envidr_composite_shell.restype = ctypes.c_int
    lib._envidr_render_bound = True


def _np32(x) -> np.ndarray:
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=np.float32)


def pack_linear(W, k_order: int, transpose: bool = False) -> np.ndarray:
    """torch-layout weight [out, in] -> MFMA tile layout (host)."""
    lib = _lib.load()
    _bind_render(lib)
    W = _np32(W)
    m_out, k_in = W.shape
    M, K = (k_in, m_out) if transpose else (m_out, k_in)
    dst = np.empty(lib.envidr_packed_weight_floats(k_order, K, M), np.float32)
    rc = lib.envidr_pack_linear(W.ctypes.data, m_out, k_in, int(transpose), k_order, dst.ctypes.data)
    if rc:
        raise _lib.EnvidrError(lib.envidr_last_error().decode())
    return dst


def pack_layer(W, bias, k_order: int, transpose: bool = False) -> np.ndarray:
    """a layer as the fused kernel streams it: [bias step] + weight steps (bias may be None)"""
    lib = _lib.load()
    _bind_render(lib)
    W = _np32(W)
    m_out, k_in = W.shape
    M, K = (k_in, m_out) if transpose else (m_out, k_in)
    b = None if bias is None else _np32(bias).reshape(-1)
    dst = np.empty(lib.envidr_packed_layer_floats(k_order, K, M, int(b is not None)), np.float32)
    rc = lib.envidr_pack_layer(W.ctypes.data, None if b is None else b.ctypes.data, m_out, k_in, int(transpose), k_order, dst.ctypes.data)
    if rc:
        raise _lib.EnvidrError(lib.envidr_last_error().decode())
    return dst


def env_orders(hidden: int, ide_dim: int = 72) -> tuple[int, int, int, int]:
    """k_order of the four environment-MLP layers (ide_dim -> hidden -> hidden -> hidden -> 12): widths with an even number of
    32-feature tiles run the hand-over form of the pass (csrc/env_pass.hip.h env_handoff) when the first layer is long enough for
    it (ide_dim / 2 - 16 >= tiles + 2 step-major steps): E1 lane order with a tile-major tail (4), E2 / E3 tile order with one (3);
    the others the plain orders.  E4 (H -> 12) is always packed for 16-row blocks (2)."""
    tiles = (hidden + 31) // 32
    return (4, 3, 3, 2) if (tiles >= 4 and tiles % 2 == 0 and ide_dim // 2 - 16 >= tiles + 2) else (0, 1, 1, 2)


ENV_MLP_SHAPES = ((72, 256), (38, 160), (72, 128), (38, 128))      # (IDE code width, hidden width) envidr_env_mlp_forward is built for


def env_mlp_supported(layers) -> bool:
    """layers: four torch.nn.Linear with bias, in -> H -> H -> H -> 12, of a shape the operator is built for"""
    if len(layers) != 4 or any(getattr(l, "bias", None) is None for l in layers):
        return False
    shapes = [tuple(l.weight.shape) for l in layers]
    k, h = shapes[0][1], shapes[0][0]
    return (k, h) in ENV_MLP_SHAPES and shapes == [(h, k), (h, h), (h, h), (12, h)] and layers[0].weight.dtype == torch.float32


def env_mlp_forward(layers, x: torch.Tensor) -> torch.Tensor:
    """The environment MLP on [..., in] IDE codes -> [..., 12] through envidr_env_mlp_forward (inference only: no autograd graph).
    The packed weights are cached on the module list and repacked when a parameter was written (its version counter) or moved."""
    lib = _lib.load()
    _bind_render(lib)
    if not x.is_cuda:
        raise _lib.EnvidrError("env_mlp_forward needs a GPU tensor; envidr_amd has no CPU path")
    key = tuple((p.data_ptr(), p._version) for l in layers for p in (l.weight, l.bias))
    cache = getattr(layers, "_envidr_env_blob", None)
    if cache is None or cache[0] != key:
        hidden = layers[0].weight.shape[0]
        flat = np.concatenate([pack_layer(l.weight, l.bias, o) for l, o in zip(layers, env_orders(hidden, layers[0].weight.shape[1]))])
        blob = torch.from_numpy(np.concatenate([flat, np.zeros((-flat.size) % 4096, np.float32)])).to(x.device)
        cache = (key, blob)
        try:
            layers._envidr_env_blob = cache
        except Exception:          # (a plain list of layers: no cache)
            pass
    blob = cache[1]
    k, h = layers[0].weight.shape[1], layers[0].weight.shape[0]
    x2 = x.detach().reshape(-1, k).to(torch.float32).contiguous()
    y = torch.empty(x2.shape[0], 12, device=x.device, dtype=torch.float32)
    rc = lib.envidr_env_mlp_forward(blob.data_ptr(), k, h, x2.data_ptr(), x2.shape[0], y.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream)
    if rc:
        raise _lib.EnvidrError(f"envidr_env_mlp_forward failed ({rc}): {lib.envidr_last_error().decode()}")
    return y.reshape(*x.shape[:-1], 12)


def linear_weight_grad(x: torch.Tensor, gy: torch.Tensor, bias: bool = True):
    """envidr_linear_weight_grad: dW [N_out, K_in] = gy^T x and db [N_out] = column sums of gy, for x [M, K_in], gy [M, N_out] on the GPU
    (fp32): the reduction over the M samples split across the chip on the fp32 matrix cores, partial sums added in a fixed order"""
    lib = _lib.load()
    _bind_render(lib)
    x = x.contiguous().float()
    gy = gy.contiguous().float()
    M, K = x.shape
    N = gy.shape[1]
    if gy.shape[0] != M or not x.is_cuda:
        raise _lib.EnvidrError("linear_weight_grad: x [M, K_in] and gy [M, N_out] on the GPU")
    dW = torch.empty(N, K, device=x.device)
    db = torch.empty(N, device=x.device) if bias else None
    nbytes = int(lib.envidr_linear_weight_grad_workspace_bytes(M, K, N))
    ws = torch.empty(max(nbytes // 4, 4), device=x.device)
    rc = lib.envidr_linear_weight_grad(x.data_ptr(), gy.data_ptr(), M, K, N, dW.data_ptr(), None if db is None else db.data_ptr(), 0,
                                       ws.data_ptr(), nbytes, torch.cuda.current_stream(x.device).cuda_stream)
    if rc:
        raise _lib.EnvidrError(f"envidr_linear_weight_grad failed ({rc}): {lib.envidr_last_error().decode()}")
    return dW, db


ROWS_PLAIN, ROWS_BIAS, ROWS_BIAS_RELU, ROWS_RELU_MASK = 0, 1, 2, 3


def linear_rows_supported(x: torch.Tensor, W: torch.Tensor) -> bool:
    """shapes / layouts envidr_linear_rows takes: fp32 GPU rows, the reduced width a multiple of 4 (16-byte row operands)"""
    return (x.is_cuda and x.dtype == torch.float32 and W.dtype == torch.float32 and x.dim() == 2 and W.dim() == 2 and x.shape[1] == W.shape[1]
            and x.shape[1] >= 4 and x.shape[1] % 4 == 0 and W.stride(0) >= 0 and W.stride(1) >= 0)


def linear_rows(x: torch.Tensor, W: torch.Tensor, bias: torch.Tensor | None = None, relu: bool = False, mask_act: torch.Tensor | None = None) -> torch.Tensor:
    """envidr_linear_rows: y [M, N] = epilogue(x [M, K] @ W[N, K]^T) on the fp32 matrix cores.  W may be any 2-D view (its two strides are
    passed on: W.t() of a row-major matrix costs no copy).  Epilogue: + bias; + bias then ReLU (`relu`); or * (mask_act > 0) (`mask_act` [M, N]:
    the ReLU gradient of the activation this product's result flows back through)."""
    lib = _lib.load()
    _bind_render(lib)
    if not linear_rows_supported(x, W):
        raise _lib.EnvidrError(f"linear_rows: unsupported operands x {tuple(x.shape)} {x.dtype} W {tuple(W.shape)} {W.dtype} strides {W.stride()}")
    # (a row-expanded operand -- stride (0, 1): the gradient of y.sum(0) -- has rows that overlap: the kernel wants ldx >= K)
    if x.stride(1) != 1 or x.stride(0) % 4 != 0 or x.stride(0) < x.shape[1] or x.data_ptr() % 16 != 0:
        x = x.contiguous()
    M, K = x.shape
    N = W.shape[0]
    if mask_act is not None:
        if bias is not None or relu or mask_act.shape != (M, N):
            raise _lib.EnvidrError("linear_rows: mask_act [M, N] excludes bias / relu")
        if mask_act.stride(1) != 1 or mask_act.stride(0) < N:
            mask_act = mask_act.contiguous()
        epi = ROWS_RELU_MASK
    elif bias is not None:
        bias = bias.contiguous()
        epi = ROWS_BIAS_RELU if relu else ROWS_BIAS
    else:
        if relu:
            raise _lib.EnvidrError("linear_rows: relu without a bias is not an epilogue of the operator")
        epi = ROWS_PLAIN
    y = torch.empty(M, N, device=x.device)
    rc = lib.envidr_linear_rows(x.data_ptr(), x.stride(0), M, K, W.data_ptr(), W.stride(0), W.stride(1), N, None if bias is None else bias.data_ptr(),
                                None if mask_act is None else mask_act.data_ptr(), 0 if mask_act is None else mask_act.stride(0), epi, y.data_ptr(), N,
                                torch.cuda.current_stream(x.device).cuda_stream)
    if rc:
        raise _lib.EnvidrError(f"envidr_linear_rows failed ({rc}): {lib.envidr_last_error().decode()}")
    return y


def pack_sdf_geometry(sdf) -> np.ndarray:
    """the SDF network 2L -> 64 -> 64 -> 15 for the 16-column geometry kernel (csrc/geo_eval16.hip.h)"""
    lib = _lib.load()
    _bind_render(lib)
    arrs = []
    for W, b in sdf:
        arrs += [_np32(W), _np32(b).reshape(-1)]
    if arrs[0].shape != (64, 32) or arrs[2].shape != (64, 64) or arrs[4].shape[1] != 64 or arrs[4].shape[0] < 15:
        raise _lib.EnvidrError("pack_sdf_geometry expects the SDF network 32 -> 64 -> 64 -> 15")
    arrs[4], arrs[5] = np.ascontiguousarray(arrs[4][:15]), np.ascontiguousarray(arrs[5][:15])
    dst = np.empty(lib.envidr_sdf_geometry_floats(), np.float32)
    rc = lib.envidr_pack_sdf_geometry(*[a.ctypes.data for a in arrs], dst.ctypes.data)
    if rc:
        raise _lib.EnvidrError(lib.envidr_last_error().decode())
    return dst


def pack_env_split(env) -> tuple[np.ndarray, np.ndarray]:
    """the four environment-MLP layers for the split-precision mode: (uint16 blob of (hi, lo) fp16 fragments in consumption
    order, zero-padded to whole LDS chunks; float32 biases as packed row-vector tiles)"""
    lib = _lib.load()
    _bind_render(lib)
    parts, biases = [], []
    for i, (W, b) in enumerate(env):
        W = _np32(W)
        order = 0 if i == 0 else 1
        dst = np.empty(lib.envidr_split_layer_halves(order, W.shape[1], W.shape[0]), np.uint16)
        rc = lib.envidr_pack_layer_split(W.ctypes.data, W.shape[0], W.shape[1], order, dst.ctypes.data)
        if rc:
            raise _lib.EnvidrError(lib.envidr_last_error().decode())
        parts.append(dst)
        biases.append(pack_rowvec(b))
    flat = np.concatenate(parts)
    chunk = lib.envidr_split_chunk_bytes() // 2
    return np.concatenate([flat, np.zeros((-flat.size) % chunk, np.uint16)]), np.concatenate(biases)


def pack_env_split2(env, ide_degree: int) -> tuple[np.ndarray, np.ndarray]:
    """the four environment-MLP layers for the fused-pair split-precision kernel (csrc/shade_split2.hip): (uint16 blob of (hi, lo) fp16
    fragments in ITS consumption order -- layers fused in pairs; float32 biases as packed row-vector tiles, as for the round-3 form)"""
    lib = _lib.load()
    _bind_render(lib)
    Ws = [_np32(W) for W, _ in env]
    hidden = Ws[0].shape[0]
    n = lib.envidr_env_split2_halves(ide_degree, hidden)
    if n == 0 or Ws[0].shape[1] != 2 * (2 ** ide_degree - 1 + ide_degree) or [w.shape for w in Ws[1:]] != [(hidden, hidden), (hidden, hidden), (12, hidden)]:
        raise _lib.EnvidrError(f"split precision is built for (ide_degree, env_hidden) = (5,256) and (4,160), not ({ide_degree},{hidden})")
    dst = np.empty(n, np.uint16)
    rc = lib.envidr_pack_env_split2(*[w.ctypes.data for w in Ws], ide_degree, hidden, dst.ctypes.data)
    if rc:
        raise _lib.EnvidrError(lib.envidr_last_error().decode())
    return dst, np.concatenate([pack_rowvec(b) for _, b in env])


def pack_rowvec(v) -> np.ndarray:
    lib = _lib.load()
    _bind_render(lib)
    v = _np32(v).reshape(-1)
    dst = np.empty(lib.envidr_packed_rowvec_floats(v.shape[0]), np.float32)
    rc = lib.envidr_pack_rowvec(v.ctypes.data, v.shape[0], dst.ctypes.data)
    if rc:
        raise _lib.EnvidrError(lib.envidr_last_error().decode())
    return dst


def _set_env_rotation(desc, radian) -> None:
    """rot_theta(radian)[:3,:3] (nerf/utils.py:48-52) into the descriptor; None switches the rotation off"""
    if radian is None:
        desc.has_env_rot = 0
        return
    c, s = math.cos(radian), math.sin(radian)
    R = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]], dtype=np.float64).astype(np.float32)
    for i, v in enumerate(R.reshape(-1)):
        desc.env_rot[i] = float(v)
    desc.has_env_rot = 1


def _shade(lib, desc, normals, dirs, geo_feat, roughness, env_rot_radian, out):
    """envidr_shade_samples on torch device tensors"""
    dev = normals.device
    normals = normals.contiguous().view(-1, 3).float()
    dirs = dirs.contiguous().view(-1, 3).float()
    M = normals.shape[0]
    geo_feat = torch.as_tensor(geo_feat, dtype=torch.float32, device=dev).contiguous()
    roughness = torch.as_tensor(roughness, dtype=torch.float32, device=dev).contiguous()
    geo_stride = 0 if geo_feat.numel() == 12 else 12
    rough_stride = 0 if roughness.numel() == 1 else 1
    if (geo_stride and geo_feat.numel() != 12 * M) or (rough_stride and roughness.numel() != M) or dirs.shape[0] != M:
        raise _lib.EnvidrError("shade: geo_feat must be [12] or [M,12], roughness a scalar or [M], dirs [M,3]")
    res = out if out is not None else {}
    for name in ("c_diffuse", "c_specular"):
        if name not in res or res[name].shape != (M, 3):
            res[name] = torch.empty(M, 3, dtype=torch.float32, device=dev)
    _set_env_rotation(desc, env_rot_radian)
    rc = lib.envidr_shade_samples(ctypes.byref(desc), normals.data_ptr(), dirs.data_ptr(), geo_feat.data_ptr(), geo_stride,
                                  roughness.data_ptr(), rough_stride, M, res["c_diffuse"].data_ptr(), res["c_specular"].data_ptr(),
                                  torch.cuda.current_stream(dev).cuda_stream)
    if rc:
        raise _lib.EnvidrError(f"envidr_shade_samples failed ({rc}): {lib.envidr_last_error().decode()}")
    return res


class FrameOverflow(_lib.EnvidrError):
    """a frame enqueued without waiting turned out not to fit its sample / record buffers (they have been grown)"""


@dataclass
class GeometryCache:
    """everything about one camera's frame that does not depend on the environment: per composited
    sample, sorted by (ray, sample index): compositing weight, normal, view direction, geometry feature, roughness;
    per ray: sample offsets and the environment-independent images"""
    n_rays: int
    offsets: torch.Tensor          # [N+1] int32
    w: torch.Tensor                # [M]
    normals: torch.Tensor          # [M,3]
    dirs: torch.Tensor             # [M,3]
    geo_feat: torch.Tensor         # [M,12]
    roughness: torch.Tensor        # [M]
    depth: torch.Tensor            # [N]
    weights_sum: torch.Tensor      # [N]
    normal_image: torch.Tensor     # [N,3]
    roughness_image: torch.Tensor  # [N]

    @property
    def n_samples(self) -> int:
        return int(self.w.shape[0])


class FusedShader:
    """Shading MLPs only (environment MLP + diffuse / specular heads) resident on the device, for samples whose
    geometry is known: surface rendering as in the reference's demo.ipynb (BASELINE configs[0]) and re-lighting.

    mlps: {"env": 4 layers, "diffuse": 2, "specular": 3} of (weight [out,in], bias [out])."""

    def __init__(self, mlps: dict, ide_degree: int = 4, diffuse_kappa_inv: float = 0.64, light_intensity_scale: float = 1.0,
                 device: str | torch.device = "cuda"):
        self.lib = _lib.load()
        _bind_render(self.lib)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.EnvidrError("FusedShader needs a GPU device; envidr_amd has no CPU path")
        env, dif, spc = mlps["env"], mlps["diffuse"], mlps["specular"]
        ide_dim = (2 ** ide_degree - 1 + ide_degree) * 2
        if len(env) != 4 or _np32(env[0][0]).shape[1] != ide_dim or _np32(dif[0][0]).shape != (32, 24) or _np32(spc[0][0]).shape != (64, 28):
            raise _lib.EnvidrError("FusedShader expects env ide_dim->H->H->H->12, diffuse 24->32->3, specular 28->64->64->3")
        self._keep = []

        def blob(parts):
            flat = np.concatenate(parts)
            t = torch.from_numpy(np.concatenate([flat, np.zeros((-flat.size) % 4096, np.float32)])).to(self.device)
            self._keep.append(t)
            return t.data_ptr()

        L = lambda Wb, order: pack_layer(Wb[0], Wb[1], order)
        d = RenderDesc()
        d.env_blob = blob([L(env[i], o) for i, o in enumerate(env_orders(*_np32(env[0][0]).shape))])
        d.head_blob = blob([L(dif[0], 0), L(dif[1], 2), L(spc[0], 0), L(spc[1], 1), L(spc[2], 2)])
        d.ide_degree, d.env_hidden = ide_degree, _np32(env[0][0]).shape[0]
        d.diffuse_kappa_inv, d.light_intensity_scale, d.intensity_scale = diffuse_kappa_inv, light_intensity_scale, 1.0
        self.desc = d

    def shade(self, normals, dirs, geo_feat, roughness, env_rot_radian: float | None = None, out: dict | None = None) -> dict:
        """normals, dirs [M,3] (unit, on the GPU); geo_feat [12] or [M,12] (unit); roughness scalar or [M].
        Returns {"c_diffuse": [M,3], "c_specular": [M,3]}."""
        return _shade(self.lib, self.desc, normals, dirs, geo_feat, roughness, env_rot_radian, out)


class FusedRenderer:
    """Device-resident scene + one-launch renderer.

    mlps: dict with keys "sdf" (3 layers), "env" (4), "diffuse" (2), "specular" (3); each a list of
    (weight [out,in], bias [out]) as numpy arrays or tensors (nn.Linear layout).  Without an "env" entry the
    model is the no-environment family (BASELINE configs[1]): diffuse 12->32->3 on geo_feat, specular
    [SH(view dir) | geo_feat | SH(normal) | n.v] -> 64 -> 64 -> 3 with SH degree `opt.dir_sh_degree`.
    """

    def __init__(self, bitfield, table, offsets, per_level_scale: float, mlps: dict, beta: float,
                 opt: FusedOptions | None = None, device: str | torch.device = "cuda"):
        self.lib = _lib.load()
        _bind_render(self.lib)
        self.opt = opt or FusedOptions()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.EnvidrError("FusedRenderer needs a GPU device; envidr_amd has no CPU path")
        dev = self.device
        self._keep: list[torch.Tensor] = []

        def up(arr, dtype=torch.float32):
            # tensors already on the device are used in place (no copy of the 48.8 MB table)
            t = arr.detach() if isinstance(arr, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(arr))
            t = t.to(device=dev, dtype=dtype).contiguous()
            self._keep.append(t)
            return t

        self.bitfield = up(bitfield, torch.uint8)
        self.table = up(table if isinstance(table, torch.Tensor) else _np32(table))
        offsets = np.asarray(offsets, dtype=np.int32)
        self.num_levels = offsets.shape[0] - 1
        if self.num_levels > MAX_LEVELS or self.table.shape[1] != 2:
            raise _lib.EnvidrError("fused renderer supports hash grids with <= 16 levels of 2 features")
        sdf, env, dif, spc = mlps["sdf"], mlps.get("env"), mlps["diffuse"], mlps["specular"]
        if len(sdf) != 3 or (env is not None and len(env) != 4) or len(dif) != 2 or len(spc) != 3:
            raise _lib.EnvidrError("fused renderer expects sdf/env/diffuse/specular MLPs of 3/4/2/3 layers")
        feat = 2 * self.num_levels
        if _np32(sdf[0][0]).shape != (64, feat) or _np32(sdf[1][0]).shape != (64, 64) or _np32(sdf[2][0]).shape[1] != 64:
            raise _lib.EnvidrError("fused renderer expects the SDF network 2L -> 64 -> 64 -> 15")
        if env is not None:
            env_hidden = _np32(env[0][0]).shape[0]
            ide_dim = (2 ** self.opt.ide_degree - 1 + self.opt.ide_degree) * 2
            if _np32(env[0][0]).shape[1] != ide_dim:
                raise _lib.EnvidrError(f"env MLP input {_np32(env[0][0]).shape[1]} != IDE dim {ide_dim} of degree {self.opt.ide_degree}")
            want_d, want_s, sh_degree = 24, 28, 0
        else:
            env_hidden, sh_degree = 0, int(self.opt.dir_sh_degree)
            if sh_degree <= 0:
                raise _lib.EnvidrError("a model without an environment MLP needs FusedOptions.dir_sh_degree (SH degree of view dir / normal)")
            want_d, want_s = 12, 2 * sh_degree ** 2 + 13
        if _np32(dif[0][0]).shape != (32, want_d) or _np32(spc[0][0]).shape != (64, want_s):
            raise _lib.EnvidrError(f"fused renderer expects diffuse {want_d}->32->3 and specular {want_s}->64->64->3 heads")

        d = RenderDesc()
        d.density_bitfield = self.bitfield.data_ptr()
        d.bound = self.opt.bound
        d.cascades = 1 + math.ceil(math.log2(self.opt.bound))
        d.grid_size = self.opt.grid_size
        d.min_near, d.max_steps, d.dt_gamma = self.opt.min_near, self.opt.max_steps, self.opt.dt_gamma
        d.T_thresh, d.density_scale, d.bg_color = self.opt.T_thresh, self.opt.density_scale, self.opt.bg_color
        d.hash_table = self.table.data_ptr()
        for i, o in enumerate(offsets):
            d.hash_offsets[i] = int(o)
        d.num_levels = self.num_levels
        d.base_resolution = self.opt.base_resolution
        d.log2_per_level_scale = float(np.log2(per_level_scale))
        d.enabled_levels = self.opt.enabled_levels

        def blob(parts):
            """concatenate packed layers in consumption order, zero-padded to whole 16 KiB LDS chunks"""
            flat = np.concatenate(parts)
            pad = (-flat.size) % 4096
            return up(np.concatenate([flat, np.zeros(pad, np.float32)])).data_ptr()

        L = lambda Wb, order: pack_layer(Wb[0], Wb[1], order)
        d.sdf_blob = blob([L(sdf[0], 0), L(sdf[1], 1), L(sdf[2], 2),
                           pack_layer(sdf[1][0], None, 1, transpose=True), pack_layer(sdf[0][0], None, 1, transpose=True)])
        d.env_blob = blob([L(env[i], o) for i, o in enumerate(env_orders(*_np32(env[0][0]).shape))]) if env is not None else None
        d.dir_sh_degree = sh_degree
        d.head_blob = blob([L(dif[0], 0), L(dif[1], 2), L(spc[0], 0), L(spc[1], 1), L(spc[2], 2)])
        renv = mlps.get("renv")
        if renv is not None and env is not None:
            if len(renv) != 4 or _np32(renv[0][0]).shape != (64, 4) or _np32(renv[3][0]).shape != (12, 64):
                raise _lib.EnvidrError("fused renderer expects the renv MLP 4 -> 64 -> 64 -> 64 -> 12")
            d.renv_blob = blob([L(renv[0], 0), L(renv[1], 1), L(renv[2], 1), L(renv[3], 2)])
            d.spec2_blob = blob([L(spc[0], 0), L(spc[1], 1), L(spc[2], 2)])
        d.indir_roughness_thresh = self.opt.indir_roughness_thresh
        d.sdf_w3_row0 = up(pack_rowvec(_np32(sdf[2][0])[0])).data_ptr()
        if self.opt.geometry_kernel not in ("32", "16"):
            raise _lib.EnvidrError(f"geometry_kernel must be '32' or '16', not {self.opt.geometry_kernel!r}")
        if self.opt.geometry_kernel == "16":
            d.sdf_geo_blob = up(pack_sdf_geometry(sdf)).data_ptr()
        d.beta = float(min(max(beta, self.opt.beta_min), self.opt.beta_max))     # LaplaceDensity.get_beta clamp
        d.roughness_bias, d.roughness_act_scale = self.opt.roughness_bias, self.opt.roughness_act_scale
        d.roughness_scale = self.opt.roughness_scale
        d.ide_degree, d.env_hidden = self.opt.ide_degree, env_hidden
        d.diffuse_kappa_inv = self.opt.diffuse_kappa_inv
        d.light_intensity_scale, d.intensity_scale = self.opt.light_intensity_scale, self.opt.intensity_scale
        d.has_env_rot = 0
        self.desc = d
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self._env_layers = env
        self._split = None            # (blob tensor, bias tensor) of the split-precision mode, packed on first use
        self._env_feat = None
        if self.opt.env_precision not in ("fp32", "f16x2", "f16x2_v1"):
            raise _lib.EnvidrError(f"env_precision must be 'fp32', 'f16x2' or 'f16x2_v1', not {self.opt.env_precision!r}")

    def _set_precision(self, precision: str | None, records: int) -> None:
        """point the descriptor at the split-precision weights + a feature scratch of `records` rows, or clear them.  "f16x2" = the fused-pair
        kernel (csrc/shade_split2.hip); "f16x2_v1" = the round-3 form (csrc/shade_split.hip), kept as its bit-for-bit cross-check"""
        precision = precision or self.opt.env_precision
        d = self.desc
        if precision == "fp32":
            d.env_split_blob = d.env_split_bias = d.env_features = None
            return
        if precision not in ("f16x2", "f16x2_v1"):
            raise _lib.EnvidrError(f"env_precision must be 'fp32', 'f16x2' or 'f16x2_v1', not {precision!r}")
        if self._env_layers is None:
            raise _lib.EnvidrError("split precision belongs to the environment-MLP family")
        form = 1 if precision == "f16x2" else 0
        if self._split is None:
            self._split = {}
        if form not in self._split:
            blob, bias = pack_env_split2(self._env_layers, self.opt.ide_degree) if form == 1 else pack_env_split(self._env_layers)
            self._split[form] = (torch.from_numpy(blob.view(np.int16)).to(self.device), torch.from_numpy(bias).to(self.device))
        if self._env_feat is None or self._env_feat.shape[0] < records:
            self._env_feat = torch.empty(max(records, 1), 24, device=self.device)
        d.env_split_blob, d.env_split_bias, d.env_features = self._split[form][0].data_ptr(), self._split[form][1].data_ptr(), self._env_feat.data_ptr()
        d.env_split_form = form

    @classmethod
    def from_scene(cls, scene, opt: FusedOptions | None = None, device="cuda"):
        """scene: envidr_amd.scenes.SceneParams"""
        opt = opt or FusedOptions(bound=scene.bound, grid_size=scene.grid_size)
        return cls(scene.bitfield, scene.table, scene.offsets, scene.per_level_scale, scene.mlps, scene.beta, opt, device)

    def set_aabb(self, aabb) -> None:
        """the box rays are intersected with (NeRFRenderer.aabb_infer); None = [-bound, bound]^3"""
        if aabb is None:
            self.desc.has_aabb = 0
            self._aabb_key = None
            return
        if isinstance(aabb, torch.Tensor):
            # the host copy of a device buffer costs a synchronisation: keep it until the tensor is replaced or written to
            key = (aabb.data_ptr(), aabb._version, aabb.device)
            if self.__dict__.get("_aabb_key") == key and self.desc.has_aabb:
                return
            vals = [float(v) for v in aabb.detach().cpu().reshape(-1).tolist()]
            self._aabb_key = key
        else:
            vals = [float(v) for v in np.asarray(aabb).reshape(-1)]
            self._aabb_key = None
        if len(vals) != 6:
            raise _lib.EnvidrError("aabb must have six values: xmin, ymin, zmin, xmax, ymax, zmax")
        for i, v in enumerate(vals):
            self.desc.aabb[i] = v
        self.desc.has_aabb = 1

    def set_env_rotation(self, radian: float | None) -> None:
        """w_r and the diffuse normal are multiplied by rot_theta(radian)[:3,:3] (renderer.py:160-172)."""
        _set_env_rotation(self.desc, radian)

    def shade(self, normals, dirs, geo_feat, roughness, env_rot_radian: float | None = None, out: dict | None = None,
              env_precision: str | None = None) -> dict:
        """shading only, for samples with known geometry (envidr_shade_samples); environment-MLP family"""
        self._set_precision(env_precision, int(normals.numel() // 3))
        try:
            return _shade(self.lib, self.desc, normals, dirs, geo_feat, roughness, env_rot_radian, out)
        finally:
            self._set_precision("fp32", 0)

    def update_sdf(self, sdf) -> None:
        """replace the SDF network's weights (same shapes): the env-sphere mode folds the material parameters of a render call into the
        first layer's bias (nerf/network.py:369-379 concatenates them to the hash features; they are constants of the call)"""
        if self.opt.geometry_kernel != "32":
            raise _lib.EnvidrError("update_sdf: only the default geometry kernel keeps its weights in sdf_blob")
        L = lambda Wb, order: pack_layer(Wb[0], Wb[1], order)
        flat = np.concatenate([L(sdf[0], 0), L(sdf[1], 1), L(sdf[2], 2),
                               pack_layer(sdf[1][0], None, 1, transpose=True), pack_layer(sdf[0][0], None, 1, transpose=True)])
        flat = np.concatenate([flat, np.zeros((-flat.size) % 4096, np.float32)])
        t = torch.from_numpy(flat).to(self.device)
        row = torch.from_numpy(pack_rowvec(_np32(sdf[2][0])[0])).to(self.device)
        # kernels already enqueued may still read the previous weights: the last few sets stay referenced (60 KiB each), and the
        # oldest is only dropped after the stream has drained
        live = self.__dict__.setdefault("_sdf_sets", [])
        live.append((t, row))
        if len(live) > 4:
            torch.cuda.current_stream(self.device).synchronize()
            del live[0]
        self.desc.sdf_blob, self.desc.sdf_w3_row0 = t.data_ptr(), row.data_ptr()

    # ---- env-sphere mode (sph_ray.py run_sph): analytic hits, S samples per hit ray, torch-formula compositing ----------
    def sphere_intersections(self, rays_o, rays_d, radius: float):
        """envidr_sphere_intersections: nears [N], fars [N], mask [N] bool of rays [N,3] against the sphere |x| = radius"""
        N, dev = int(rays_o.shape[0]), rays_o.device
        nears, fars, mask = torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(N, dtype=torch.uint8, device=dev)
        rc = self.lib.envidr_sphere_intersections(rays_o.data_ptr(), rays_d.data_ptr(), N, float(radius), nears.data_ptr(), fars.data_ptr(),
                                                  mask.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        if rc:
            raise _lib.EnvidrError(f"envidr_sphere_intersections failed ({rc}): {self.lib.envidr_last_error().decode()}")
        return nears, fars, mask.bool()

    def shell_samples(self, rays_o, rays_d, hit_rays, nears, z_offsets, step_size: float, noise=None):
        """envidr_shell_samples: sample-major xyz [S,M,3], dirs [S,M,3], z_vals [S,M] of the hit rays"""
        M, S, dev = int(hit_rays.shape[0]), int(z_offsets.shape[0]), rays_o.device
        xyz, dirs, z = torch.empty(S, M, 3, device=dev), torch.empty(S, M, 3, device=dev), torch.empty(S, M, device=dev)
        rc = self.lib.envidr_shell_samples(rays_o.data_ptr(), rays_d.data_ptr(), hit_rays.data_ptr(), nears.data_ptr(), z_offsets.data_ptr(),
                                           None if noise is None else noise.contiguous().float().data_ptr(), float(step_size), M, S,
                                           xyz.data_ptr(), dirs.data_ptr(), z.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        if rc:
            raise _lib.EnvidrError(f"envidr_shell_samples failed ({rc}): {self.lib.envidr_last_error().decode()}")
        return xyz, dirs, z

    def composite_shell(self, sigma, z_vals, c_diffuse, c_specular, normals, roughness, hit_slot, nears, far_max, bg, step_size: float,
                        want=("normal_image", "diffuse_image", "specular_image", "roughness_image")) -> dict:
        """envidr_composite_shell over all N rays (hit_slot [N] int32: -1 = the ray misses the sphere)"""
        N, dev = int(hit_slot.shape[0]), hit_slot.device
        S, M = int(z_vals.shape[0]), int(z_vals.shape[1])
        shapes = {"image": (3,), "depth": (), "weights_sum": (), "normal_image": (3,), "diffuse_image": (3,), "specular_image": (3,),
                  "roughness_image": ()}
        out = {k: torch.empty(N, *shapes[k], device=dev) for k in ("image", "depth", "weights_sum", *want)}
        ptr = lambda k: out[k].data_ptr() if k in out else None
        opt = lambda t: None if t is None else t.data_ptr()
        rc = self.lib.envidr_composite_shell(sigma.data_ptr(), z_vals.data_ptr(), c_diffuse.data_ptr(), c_specular.data_ptr(), opt(normals),
                                             opt(roughness), hit_slot.data_ptr(), nears.data_ptr(), far_max.data_ptr(), bg.data_ptr(), N, M, S,
                                             float(step_size), float(self.desc.intensity_scale), ptr("image"), ptr("depth"), ptr("weights_sum"),
                                             ptr("normal_image"), ptr("diffuse_image"), ptr("specular_image"), ptr("roughness_image"),
                                             torch.cuda.current_stream(dev).cuda_stream)
        if rc:
            raise _lib.EnvidrError(f"envidr_composite_shell failed ({rc}): {self.lib.envidr_last_error().decode()}")
        return out

    # ---- geometry cache: march / hash / SDF once per camera, shading per environment -----------------
    def cache_geometry(self, rays_o: torch.Tensor, rays_d: torch.Tensor, samples_per_ray_hint: float = 16.0) -> GeometryCache:
        """one geometry-only render that also exports every composited sample; retried once with the exact size if the
        first guess of the record capacity was too small"""
        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        N, dev = rays_o.shape[0], rays_o.device
        cap = max(int(N * samples_per_ray_hint), 1024)
        for _ in range(2):
            counter = torch.zeros(1, dtype=torch.int32, device=dev)
            rec = {"ray": torch.empty(cap, dtype=torch.int32, device=dev), "idx": torch.empty(cap, dtype=torch.int32, device=dev),
                   "w": torch.empty(cap, device=dev), "normal": torch.empty(cap, 3, device=dev),
                   "geo": torch.empty(cap, 12, device=dev), "rough": torch.empty(cap, device=dev)}
            ex = GeometryExport(counter.data_ptr(), cap, rec["ray"].data_ptr(), rec["idx"].data_ptr(), rec["w"].data_ptr(),
                                rec["normal"].data_ptr(), rec["geo"].data_ptr(), rec["rough"].data_ptr())
            self.desc.geometry_export = ctypes.pointer(ex)
            try:
                res = self.render(rays_o, rays_d, None, extras=True, geometry_only=True)
            finally:
                self.desc.geometry_export = None
            M = int(counter.item())
            if M <= cap:
                break
            cap = M
        else:
            raise _lib.EnvidrError("geometry export: record count changed between two identical renders")
        ray = rec["ray"][:M].long()
        order = torch.argsort(ray * 4096 + rec["idx"][:M].long())          # (ray, sample index): at most 1024 samples per ray
        ray = ray[order]
        offsets = torch.zeros(N + 1, dtype=torch.int32, device=dev)
        offsets[1:] = torch.cumsum(torch.bincount(ray, minlength=N), 0).int()
        return GeometryCache(n_rays=N, offsets=offsets, w=rec["w"][:M][order].contiguous(),
                             normals=rec["normal"][:M][order].contiguous(), dirs=rays_d[ray].contiguous(),
                             geo_feat=rec["geo"][:M][order].contiguous(), roughness=rec["rough"][:M][order].contiguous(),
                             depth=res["depth"].clone(), weights_sum=res["weights_sum"].clone(),
                             normal_image=res["normal_image"].clone(), roughness_image=None)

    # ---- two-phase frame: geometry pass -> shading pass, everything recomputed every frame ---------------------
    def render_two_phase(self, rays_o: torch.Tensor, rays_d: torch.Tensor, env_rot_radian: float | None = None,
                         out: dict | None = None, ray_cost: torch.Tensor | None = None, samples_per_ray_hint: float = 24.0,
                         events: list | None = None) -> dict:
        """The same frame as render() (bit-identical outputs, environment-MLP family), scheduled as two passes:
          1. geometry: march + hash grid + SDF network + normals + compositing weights for every sample, one record per
             composited sample appended to device buffers (a geometry_only launch of the persistent kernel);
          2. shading: the records streamed through IDE + environment MLP x2 + heads (k_shade_samples: nothing but dense
             layers, every lane busy every round), then composited per ray.
        The per-sample records cross HBM once (80 B per sample); in exchange the shading kernel has no ray tail, no
        divergent marching and no idle lanes.  `ray_cost` as in render().  Buffers are kept between calls.
        `events`: four torch.cuda.Event(enable_timing=True) recorded at the pass boundaries (geometry | shading | composite)."""
        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        N, dev = rays_o.shape[0], rays_o.device
        res = out if out is not None else {}
        if N == 0:                      # no rays: an empty result, nothing enqueued
            for name, shape in (("image", (0, 3)), ("depth", (0,)), ("weights_sum", (0,)), ("normal_image", (0, 3)), ("diffuse_image", (0, 3)),
                                ("specular_image", (0, 3)), ("roughness_image", (0,))):
                res[name] = torch.empty(*shape, device=dev)
            res["n_records"] = 0
            if events:
                for e in events:
                    e.record()
            return res
        st = self.__dict__.setdefault("_two_phase", {})
        if ray_cost is None:
            if st.get("cost_n") != N:
                st["cost"], st["cost_n"] = torch.zeros(N, dtype=torch.int16, device=dev), N
            ray_cost = st["cost"]
        cap = st.get("cap", 0)
        if cap < 1024 or st.get("cap_n") != N:
            cap = max(int(N * samples_per_ray_hint), 1024)
        stream = torch.cuda.current_stream(dev).cuda_stream
        ev = events
        for attempt in range(2):
            if st.get("cap") != cap or st.get("cap_n") != N:
                st.update(cap=cap, cap_n=N, counter=torch.zeros(1, dtype=torch.int32, device=dev),
                          ray=torch.empty(cap, dtype=torch.int32, device=dev), idx=torch.empty(cap, dtype=torch.int32, device=dev),
                          w=torch.empty(cap, device=dev), normal=torch.empty(cap, 3, device=dev), geo=torch.empty(cap, 12, device=dev),
                          rough=torch.empty(cap, device=dev), perm=torch.empty(cap, dtype=torch.int32, device=dev),
                          cd=torch.empty(cap, 3, device=dev), cs=torch.empty(cap, 3, device=dev))
            ex = GeometryExport(st["counter"].data_ptr(), cap, st["ray"].data_ptr(), st["idx"].data_ptr(), st["w"].data_ptr(),
                                st["normal"].data_ptr(), st["geo"].data_ptr(), st["rough"].data_ptr())
            st["counter"].zero_()
            if ev: ev[0].record()
            self.desc.geometry_export = ctypes.pointer(ex)
            try:
                self.render(rays_o, rays_d, None, extras=True, geometry_only=True, out=res, ray_cost=ray_cost)
            finally:
                self.desc.geometry_export = None
            if ev: ev[1].record()
            # shading is enqueued without waiting for the count (the kernel reads it on the device) ...
            _set_env_rotation(self.desc, env_rot_radian)
            rc = self.lib.envidr_shade_records(ctypes.byref(self.desc), ctypes.byref(ex), rays_d.data_ptr(), st["cd"].data_ptr(),
                                               st["cs"].data_ptr(), stream)
            if rc:
                raise _lib.EnvidrError(f"envidr_shade_records failed ({rc}): {self.lib.envidr_last_error().decode()}")
            if ev: ev[2].record()
            offsets = torch.zeros(N + 1, dtype=torch.int32, device=dev)
            torch.cumsum(ray_cost, 0, dtype=torch.int32, out=offsets[1:])
            for name in ("image", "diffuse_image", "specular_image"):
                if name not in res or res[name].shape != (N, 3):
                    res[name] = torch.empty(N, 3, device=dev)
            rc = self.lib.envidr_composite_records(ctypes.byref(ex), offsets.data_ptr(), st["perm"].data_ptr(), st["cd"].data_ptr(),
                                                   st["cs"].data_ptr(), res["weights_sum"].data_ptr(), N,
                                                   float(self.desc.intensity_scale), float(self.desc.bg_color), res["image"].data_ptr(),
                                                   res["diffuse_image"].data_ptr(), res["specular_image"].data_ptr(), stream)
            if rc:
                raise _lib.EnvidrError(f"envidr_composite_records failed ({rc}): {self.lib.envidr_last_error().decode()}")
            if ev: ev[3].record()
            # ... and only now is the count looked at: a frame that did not fit is redone with buffers of the right size
            M = int(st["counter"].item())
            if M <= cap:
                break
            cap = M + M // 8
        else:
            raise _lib.EnvidrError("two-phase render: record count changed between two identical geometry passes")
        res["n_records"] = M
        return res

    # ---- two-phase frame on the geometry pipeline: march rounds + per-sample evaluation -> shading -> composite ----------
    def _march_key(self) -> tuple:
        """the descriptor fields that decide which samples a frame's records are (run_cuda rewrites max_steps / T_thresh / dt_gamma per
        call, set_aabb the box): records are only shaded again by a frame that would have marched the same samples"""
        d = self.desc
        return (int(d.max_steps), float(d.T_thresh), float(d.dt_gamma), float(d.min_near), int(d.has_aabb), tuple(float(v) for v in d.aabb),
                float(d.density_scale), float(d.bound))

    def geometry_eval(self, xyz: torch.Tensor, dt: torch.Tensor | None = None, want=("sigma", "normal", "geo_feat", "roughness")) -> dict:
        """envidr_geometry_eval: hash grid + SDF network forward / input gradient + per-sample terms for positions [M,3]"""
        xyz = xyz.contiguous().view(-1, 3).float()
        M, dev = xyz.shape[0], xyz.device
        shapes = {"alpha": (), "sigma": (), "normal": (3,), "geo_feat": (12,), "roughness": (), "blend": ()}
        out = {k: torch.empty(M, *shapes[k], device=dev) for k in want}
        if "alpha" in out and dt is None:
            raise _lib.EnvidrError("geometry_eval: alpha needs the step sizes dt")
        so = SamplesOut(**{k: v.data_ptr() for k, v in out.items()})
        rc = self.lib.envidr_geometry_eval(ctypes.byref(self.desc), xyz.data_ptr(), None if dt is None else dt.contiguous().float().data_ptr(),
                                           M, None, ctypes.byref(so), torch.cuda.current_stream(dev).cuda_stream)
        if rc:
            raise _lib.EnvidrError(f"envidr_geometry_eval failed ({rc}): {self.lib.envidr_last_error().decode()}")
        return out

    def geometry_probe(self, xyz: torch.Tensor, want=("features", "corner_rows", "raw_outputs", "sdf_gradient")) -> dict:
        """envidr_geometry_probe (test hook): what the hash section and the SDF network of the frames' geometry kernel compute for
        positions [M,3]: features [M,32], corner_rows [M,16,8] int32 (bit pattern of uint32), raw_outputs [M,16], sdf_gradient [M,3]"""
        xyz = xyz.contiguous().view(-1, 3).float()
        M, dev = xyz.shape[0], xyz.device
        spec = {"features": ((M, 32), torch.float32), "corner_rows": ((M, 16, 8), torch.int32), "raw_outputs": ((M, 16), torch.float32),
                "sdf_gradient": ((M, 3), torch.float32)}
        out = {k: torch.zeros(*spec[k][0], dtype=spec[k][1], device=dev) for k in want}
        ptr = lambda k: out[k].data_ptr() if k in out else None
        rc = self.lib.envidr_geometry_probe(ctypes.byref(self.desc), xyz.data_ptr(), M, ptr("features"), ptr("corner_rows"), ptr("raw_outputs"),
                                            ptr("sdf_gradient"), torch.cuda.current_stream(dev).cuda_stream)
        if rc:
            raise _lib.EnvidrError(f"envidr_geometry_probe failed ({rc}): {self.lib.envidr_last_error().decode()}")
        return out

    def _frame_buffers(self, N: int, dev, samples_per_ray: float, buffers: str = "") -> dict:
        """workspace, record arrays, per-ray count hint and status words of the frames with N rays (a few ray counts are
        kept: the three passes of indirect rendering alternate between theirs).  `buffers` names a separate set for the same N:
        the reflected pass of an indirect frame must not overwrite the primary rays' records, which the main pass shades again."""
        frames = self.__dict__.setdefault("_frames", {})
        key = (N, buffers) if buffers else N
        st = frames.get(key)
        cap = max(int(N * samples_per_ray), 4096)
        cap = min(cap, N * int(self.desc.max_steps) + 4096)          # a ray never marches more than max_steps samples
        # trim: buffers more than twice the PEAK any frame of this ray count has needed so far (evaluated samples or records,
        # + 25 %) are given back after 64 frames and re-made at that size -- the first guess of 20 samples per ray is ~0.9 GB at
        # 800^2, harmless on 288 GB, but a long-lived renderer should converge on what it uses.  The peak never decreases, so a
        # trimmed buffer still fits every frame seen (the passes of an indirect frame share N with different needs); the per-ray
        # count hints are carried over.
        carry = None
        if st is not None and samples_per_ray <= st["hint"]:
            cap = min(cap, st["cap"])              # the guess that sized (or a trim that re-sized) these buffers does not grow them again
        if st is not None and st.get("peak") and not st["pending"] and st["cap"] >= cap:
            need = int(1.25 * st["peak"]) + 4096
            st["slack"] = st.get("slack", 0) + 1 if st["cap"] > 2 * need else 0
            if st["slack"] >= 64:
                carry = (st["costs"], st["peak"])
                frames.pop(key)
                st, cap = None, need
                # the growth hint of an earlier overflow decays with the buffers it grew: what these rays need is now known
                self.__dict__.get("_frame_hints", {}).pop(N, None)
        if st is None or st["cap"] < cap:
            if st is None and len(frames) >= 6:
                frames.pop(next(iter(frames)))
            need = int(self.lib.envidr_geometry_workspace_bytes(N, cap))
            try:
                st = dict(N=N, cap=cap, ws=torch.empty(need, dtype=torch.uint8, device=dev), counter=torch.zeros(1, dtype=torch.int32, device=dev),
                          ray=torch.empty(cap, dtype=torch.int32, device=dev), idx=torch.empty(cap, dtype=torch.int32, device=dev),
                          w=torch.empty(cap, device=dev), slot=torch.empty(cap, dtype=torch.int32, device=dev),
                          perm=torch.empty(cap, dtype=torch.int32, device=dev), cd=torch.empty(cap, 3, device=dev), cs=torch.empty(cap, 3, device=dev),
                          shade_list=torch.empty(cap + cap // 1024 + 3, dtype=torch.int32, device=dev),
                          cost=torch.zeros(N, dtype=torch.int16, device=dev), costs={}, offsets=torch.zeros(N + 1, dtype=torch.int32, device=dev),
                          stats=torch.zeros(3, dtype=torch.int64, device=dev), worst=torch.zeros(3, dtype=torch.int64, device=dev),
                          host=torch.zeros(6, dtype=torch.int64).pin_memory(),
                          event=torch.cuda.Event(), pending=False, hint=samples_per_ray, peak=0)
            except torch.OutOfMemoryError as e:
                # (regrowth after an overflow quadruples the capacity up to N * max_steps: on a small device that can end here)
                raise _lib.EnvidrError(f"render_frame: the frame buffers for {N} rays x {cap / max(N, 1):.1f} samples per ray ({cap} sample / record "
                                       f"slots) need {(need + 44 * cap) / 2 ** 30:.2f} GiB of device memory, which is not available; render fewer "
                                       "rays per call (opt.max_ray_batch_cuda)") from e
            if carry is not None:
                st["costs"], st["peak"] = carry
            frames[key] = st
        self.__dict__["_frame"] = st
        return st

    def check_frames(self, block: bool = True) -> None:
        """Frames are enqueued without waiting for the device.  This looks at the status words of the frames not looked at
        yet (they were copied to pinned memory behind each frame) and raises FrameOverflow if one did not fit its buffers --
        which are then dropped, so that redoing the frame allocates larger ones.  Called (block=False: only frames the device
        has finished, so that the host can enqueue a frame ahead of the device) at the start of every frame, and with
        block=True by anyone who needs the answer now."""
        overflowed = None
        for key, st in list(self.__dict__.get("_frames", {}).items()):
            N = st["N"]
            if not st["pending"]:
                continue
            if not block and not st["event"].query():
                continue                                   # still running: its (sticky) status is looked at later
            st["event"].synchronize()
            st["pending"] = False
            samples, records, _, w_samples, w_records, overflow = (int(v) for v in st["host"])
            st["last"] = (samples, records)
            st["worst"].zero_()
            samples, records = max(samples, w_samples), max(records, w_records)
            if not overflow:
                st["peak"] = max(st.get("peak", 0), samples, records)
            if overflow:
                # (the device stops counting where a frame stops fitting, so the need is only known to exceed what was seen: grow
                #  geometrically; a ray never has more than max_steps samples, which bounds the search)
                self.__dict__.setdefault("_frame_hints", {})[N] = max(4.0 * st["cap"] / N, 1.5 * max(samples, records) / N)
                del self.__dict__["_frames"][key]
                overflowed = st["cap"]
        if overflowed is not None:
            raise FrameOverflow(f"a frame needed more than the {overflowed} sample / record slots it was given; its buffers were dropped, render it again")

    def render_frame(self, rays_o: torch.Tensor, rays_d: torch.Tensor, env_rot_radian: float | None = None, out: dict | None = None,
                     samples_per_ray_hint: float = 20.0, events: list | None = None, geometry_only: bool = False, wait: bool = True,
                     use_cost_hint: bool = True, r_images: torch.Tensor | None = None, ray_mask: torch.Tensor | None = None,
                     tag: str = "", env_precision: str | None = None, image_width: int = 0, buffers: str = "",
                     reuse_geometry: dict | None = None) -> dict:
        """One frame as: geometry pipeline (envidr_geometry_pass: march rounds + per-sample hash grid / SDF network, one record
        per composited sample) -> shading of the records (k_shade_samples) -> per-ray composite.  Both network families
        (environment MLP; SH heads without one); `r_images` [N,4] (reflected radiance + visibility per ray) selects the
        reflected-radiance branch of the third pass of indirect rendering; `geometry_only` is its first pass.  `ray_mask`
        (bool / uint8 [N]): rays with 0 are not rendered (background, zero weight).  `tag` separates the per-ray count hints
        of different uses of the same N rays (the three passes of an indirect frame march different rays).
        The per-ray sample counts of the previous frame of these N rays (kept in the frame buffers) size each ray's first march
        chunk (use_cost_hint; results do not depend on it).  Nothing waits for the device while the frame is enqueued; with wait=False the call does not wait at the end either
        (video loops: check_frames() -- called by the next frame -- raises FrameOverflow if a frame did not fit).
        `events`: four torch.cuda.Event(enable_timing=True) recorded at the pass boundaries (geometry | shading | composite).
        `env_precision`: "fp32" / "f16x2" for this frame (default: FusedOptions.env_precision).
        `image_width`: the rays are the pixels of a row-major image this wide (layout hint: blocks of 64 rays are then 8x8-pixel
        tiles; same outputs, bit for bit; ignored unless width and height are multiples of 8).
        `buffers`: name of the buffer set (default: one per N).
        `reuse_geometry`: the result of the previous render_frame of THESE rays in THIS buffer set (a geometry_only frame): its
        records are shaded and composited again -- no marching, no hash grid, no SDF network; `ray_mask` then zeroes the
        masked-out rays' outputs.  Main pass of indirect rendering: same rays, same samples as the first pass."""
        self.check_frames(block=False)
        if not (rays_o.is_cuda and rays_d.is_cuda):
            raise _lib.EnvidrError("render_frame: rays_o / rays_d must be CUDA tensors (envidr_amd has no CPU path)")      # the reference's CHECK_CUDA
        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        N, dev = rays_o.shape[0], rays_o.device
        res = out if out is not None else {}
        if N == 0:
            # no rays (a rank whose tile shard is empty, a masked batch that came out empty): an empty result, nothing enqueued
            for name, shape in (("depth", (0,)), ("weights_sum", (0,)), ("normal_image", (0, 3)), ("roughness_image", (0,)), ("image", (0, 3)),
                                ("diffuse_image", (0, 3)), ("specular_image", (0, 3))):
                if geometry_only and name in ("diffuse_image", "specular_image"):
                    continue
                res[name] = torch.empty(*shape, device=dev)
            res["ray_cost"] = torch.empty(0, dtype=torch.int16, device=dev)
            res["n_records"] = res["n_samples"] = 0
            if events:
                for e in events:
                    e.record()
            return res
        if reuse_geometry is not None:
            st = self.__dict__.get("_frames", {}).get((N, buffers) if buffers else N)
            if (geometry_only or st is None or st.get("records_of") != (rays_o.data_ptr(), rays_d.data_ptr(), N, self._march_key())
                    or reuse_geometry.get("_records_serial") != st.get("records_serial")):
                reuse_geometry = None                      # not the frame these buffers hold: render from scratch
        if reuse_geometry is not None:
            return self._reshade_records(st, reuse_geometry, rays_d, N, dev, env_rot_radian, res, events, r_images, ray_mask, env_precision, tag)
        for attempt in range(8):          # capacities 20, 80, 320, 1280 ... samples per ray: max_steps (<= 65535) is reached within eight
            st = self._frame_buffers(N, dev, max(samples_per_ray_hint, self.__dict__.get("_frame_hints", {}).get(N, 0.0)), buffers)
            cap = st["cap"]
            stream = torch.cuda.current_stream(dev).cuda_stream
            for name, shape in (("depth", (N,)), ("weights_sum", (N,)), ("normal_image", (N, 3)), ("roughness_image", (N,))):
                if name not in res or res[name].shape != shape:
                    res[name] = torch.empty(*shape, device=dev)
            o = RenderOut()
            o.depth, o.weights_sum = res["depth"].data_ptr(), res["weights_sum"].data_ptr()
            o.normal_image, o.roughness_image = res["normal_image"].data_ptr(), res["roughness_image"].data_ptr()
            o.stats = st["stats"].data_ptr()
            ex = GeometryExport(st["counter"].data_ptr(), cap, st["ray"].data_ptr(), st["idx"].data_ptr(), st["w"].data_ptr(), None, None, None,
                                st["slot"].data_ptr(), None)
            if self.opt.skip_zero_weight:          # records whose compositing weight is exactly 0 are not shaded (same outputs)
                ex.shade_list = st["shade_list"].data_ptr()
            if tag not in st["costs"]:
                st["costs"][tag] = torch.zeros(N, dtype=torch.int16, device=dev)
            st["cost"] = st["costs"][tag]
            if not use_cost_hint:
                st["cost"].zero_()
            self.desc.ray_cost = st["cost"].data_ptr()
            if ray_mask is not None:
                ray_mask = ray_mask.reshape(-1).to(torch.uint8).contiguous()
                if ray_mask.shape[0] != N or not ray_mask.is_cuda:
                    raise _lib.EnvidrError("render_frame: ray_mask must be [N] on the GPU")
            self.desc.ray_mask = None if ray_mask is None else ray_mask.data_ptr()
            self.desc.image_width = int(image_width)
            self.desc.geometry_only, self.desc.r_images, self.desc.geometry_export = 0, None, None
            ev = events
            if ev: ev[0].record()
            rc = self.lib.envidr_geometry_pass(ctypes.byref(self.desc), rays_o.data_ptr(), rays_d.data_ptr(), N, ctypes.byref(o), ctypes.byref(ex),
                                               st["ws"].data_ptr(), st["ws"].numel(), cap, stream)
            self.desc.ray_cost = None
            self.desc.ray_mask = None
            self.desc.image_width = 0
            if rc:
                raise _lib.EnvidrError(f"envidr_geometry_pass failed ({rc}): {self.lib.envidr_last_error().decode()}")
            if ev: ev[1].record()
            if not geometry_only:
                _set_env_rotation(self.desc, env_rot_radian)
                if r_images is not None:
                    if not self.desc.renv_blob:
                        raise _lib.EnvidrError("render_frame: r_images given but the model has no renv MLP")
                    r_images = r_images.contiguous().view(-1, 4).float()
                    if r_images.shape[0] != N or not r_images.is_cuda:
                        raise _lib.EnvidrError("render_frame: r_images must be [N,4] on the GPU")
                    self.desc.r_images = r_images.data_ptr()
                self._set_precision("fp32" if r_images is not None or self._env_layers is None else env_precision, st["cap"])
                rc = self.lib.envidr_shade_records(ctypes.byref(self.desc), ctypes.byref(ex), rays_d.data_ptr(), st["cd"].data_ptr(),
                                                   st["cs"].data_ptr(), stream)
                self.desc.r_images = None
                self._set_precision("fp32", 0)
                if rc:
                    raise _lib.EnvidrError(f"envidr_shade_records failed ({rc}): {self.lib.envidr_last_error().decode()}")
            if ev: ev[2].record()
            if not geometry_only:
                torch.cumsum(st["cost"], 0, dtype=torch.int32, out=st["offsets"][1:])
                for name in ("image", "diffuse_image", "specular_image"):
                    if name not in res or res[name].shape != (N, 3):
                        res[name] = torch.empty(N, 3, device=dev)
                rc = self.lib.envidr_composite_records(ctypes.byref(ex), st["offsets"].data_ptr(), st["perm"].data_ptr(), st["cd"].data_ptr(),
                                                       st["cs"].data_ptr(), res["weights_sum"].data_ptr(), N, float(self.desc.intensity_scale),
                                                       float(self.desc.bg_color), res["image"].data_ptr(), res["diffuse_image"].data_ptr(),
                                                       res["specular_image"].data_ptr(), stream)
                if rc:
                    raise _lib.EnvidrError(f"envidr_composite_records failed ({rc}): {self.lib.envidr_last_error().decode()}")
            else:
                res["image"] = ((1 - res["weights_sum"]) * float(self.desc.bg_color))[:, None].expand(N, 3).contiguous()
            if ev: ev[3].record()
            # status of this frame + the worst since the last look (several frames may be in flight: the three passes of an
            # indirect frame, a video loop)
            torch.maximum(st["worst"], st["stats"], out=st["worst"])
            st["host"][:3].copy_(st["stats"], non_blocking=True)
            st["host"][3:].copy_(st["worst"], non_blocking=True)
            st["event"].record()
            st["pending"] = True
            if self.__dict__.get("frame_log") is not None:       # measurement hook: (samples evaluated, records, overflow) per tag, on the device
                self.frame_log[tag] = st["stats"].clone()
            res["ray_cost"] = st["cost"]
            # what these buffers hold now (reuse_geometry asks for exactly this frame's records)
            st["records_of"] = (rays_o.data_ptr(), rays_d.data_ptr(), N, self._march_key())
            st["records_serial"] = res["_records_serial"] = st.get("records_serial", 0) + 1
            st["records_cost"], st["records_ex"] = st["cost"], ex       # (envidr_geometry_pass filled in where the per-sample arrays live)
            if not wait:
                return res
            try:
                self.check_frames()
            except FrameOverflow:
                continue
            res["n_records"] = st["last"][1]
            res["n_samples"] = st["last"][0]
            return res
        raise _lib.EnvidrError("render_frame: the frame did not fit its buffers after seven enlargements")

    def _reshade_records(self, st, prev, rays_d, N, dev, env_rot_radian, res, events, r_images, ray_mask, env_precision, tag) -> dict:
        """shading + composite over the records a geometry frame left in `st` (render_frame's reuse_geometry): the per-ray outputs
        of that frame are taken over (masked), every record is shaded with this call's environment / reflected radiance"""
        stream = torch.cuda.current_stream(dev).cuda_stream
        ev = events
        if ev: ev[0].record()
        keep = None
        if ray_mask is not None:
            keep = ray_mask.reshape(-1).to(torch.float32)
            if keep.shape[0] != N or not keep.is_cuda:
                raise _lib.EnvidrError("render_frame: ray_mask must be [N] on the GPU")
        for name in ("depth", "weights_sum", "roughness_image"):
            res[name] = prev[name] if keep is None else prev[name] * keep
        res["normal_image"] = prev["normal_image"] if keep is None else prev["normal_image"] * keep[:, None]
        if ev: ev[1].record()
        ex = st["records_ex"]
        _set_env_rotation(self.desc, env_rot_radian)
        self.desc.geometry_only, self.desc.r_images, self.desc.geometry_export = 0, None, None
        if r_images is not None:
            if not self.desc.renv_blob:
                raise _lib.EnvidrError("render_frame: r_images given but the model has no renv MLP")
            r_images = r_images.contiguous().view(-1, 4).float()
            if r_images.shape[0] != N or not r_images.is_cuda:
                raise _lib.EnvidrError("render_frame: r_images must be [N,4] on the GPU")
            self.desc.r_images = r_images.data_ptr()
        self._set_precision("fp32" if r_images is not None or self._env_layers is None else env_precision, st["cap"])
        rc = self.lib.envidr_shade_records(ctypes.byref(self.desc), ctypes.byref(ex), rays_d.data_ptr(), st["cd"].data_ptr(), st["cs"].data_ptr(), stream)
        self.desc.r_images = None
        self._set_precision("fp32", 0)
        if rc:
            raise _lib.EnvidrError(f"envidr_shade_records failed ({rc}): {self.lib.envidr_last_error().decode()}")
        if ev: ev[2].record()
        torch.cumsum(st["records_cost"], 0, dtype=torch.int32, out=st["offsets"][1:])
        for name in ("image", "diffuse_image", "specular_image"):
            if name not in res or res[name].shape != (N, 3):
                res[name] = torch.empty(N, 3, device=dev)
        # (composited with the UNMASKED weights: a masked-out ray's pixel is then replaced below, everybody else's is what a
        #  from-scratch frame with this mask computes, bit for bit)
        rc = self.lib.envidr_composite_records(ctypes.byref(ex), st["offsets"].data_ptr(), st["perm"].data_ptr(), st["cd"].data_ptr(),
                                               st["cs"].data_ptr(), prev["weights_sum"].data_ptr(), N, float(self.desc.intensity_scale),
                                               float(self.desc.bg_color), res["image"].data_ptr(), res["diffuse_image"].data_ptr(),
                                               res["specular_image"].data_ptr(), stream)
        if rc:
            raise _lib.EnvidrError(f"envidr_composite_records failed ({rc}): {self.lib.envidr_last_error().decode()}")
        if keep is not None:
            k3 = keep[:, None]
            res["image"] = res["image"] * k3 + (1 - k3) * float(self.desc.bg_color)
            res["diffuse_image"] = res["diffuse_image"] * k3
            res["specular_image"] = res["specular_image"] * k3
        if ev: ev[3].record()
        if self.__dict__.get("frame_log") is not None:       # (samples evaluated: none; records shaded: the geometry frame's; overflow: its)
            self.frame_log[tag] = st["stats"] * torch.tensor([0, 1, 1], dtype=st["stats"].dtype, device=dev)
        res["ray_cost"] = st["records_cost"]
        return res

    def render_cached(self, cache: GeometryCache, env_rot_radian: float | None = None, out: dict | None = None) -> dict:
        """re-light a cached frame: envidr_shade_samples over its samples + envidr_composite_shaded; bit-identical to
        render() of the same rays with the same environment rotation"""
        dev = cache.w.device
        res = out if out is not None else {}
        N = cache.n_rays
        for name in ("image", "diffuse_image", "specular_image"):
            if name not in res or res[name].shape != (N, 3):
                res[name] = torch.empty(N, 3, device=dev)
        if cache.w.shape[0] == 0:       # no ray of this camera hits anything (or no rays): background only, nothing to shade
            res["image"].fill_(float(self.desc.bg_color))
            res["diffuse_image"].zero_()
            res["specular_image"].zero_()
            res.update(depth=cache.depth, weights_sum=cache.weights_sum, normal_image=cache.normal_image)
            return res
        shaded = _shade(self.lib, self.desc, cache.normals, cache.dirs, cache.geo_feat, cache.roughness, env_rot_radian,
                        res.setdefault("_shaded", {}))
        rc = self.lib.envidr_composite_shaded(cache.offsets.data_ptr(), cache.w.data_ptr(), shaded["c_diffuse"].data_ptr(),
                                              shaded["c_specular"].data_ptr(), cache.weights_sum.data_ptr(), N,
                                              float(self.desc.intensity_scale), float(self.desc.bg_color), res["image"].data_ptr(),
                                              res["diffuse_image"].data_ptr(), res["specular_image"].data_ptr(),
                                              torch.cuda.current_stream(dev).cuda_stream)
        if rc:
            raise _lib.EnvidrError(f"envidr_composite_shaded failed ({rc}): {self.lib.envidr_last_error().decode()}")
        res.update(depth=cache.depth, weights_sum=cache.weights_sum, normal_image=cache.normal_image)
        return res

    def render(self, rays_o: torch.Tensor, rays_d: torch.Tensor, env_rot_radian: float | None = None,
               extras: bool = True, stats: bool = False, out: dict | None = None, geometry_only: bool = False,
               r_images: torch.Tensor | None = None, ray_cost: torch.Tensor | None = None) -> dict:
        """rays_o, rays_d: [N,3] fp32 on the GPU.  Returns image [N,3], depth [N], weights_sum [N]
        (+ normal_image, diffuse_image, specular_image, roughness_image when `extras`).
        geometry_only: depth / weights_sum / normal_image only (first pass of indirect rendering).
        r_images: [N,4] reflected radiance + visibility per ray (third pass of indirect rendering).
        ray_cost: int16 [N] scheduling hint kept by the caller between renders of the same rays (zeros at first): samples
        per ray of the previous render in, of this render out; orders the work list longest ray first (same outputs)."""
        if not (rays_o.is_cuda and rays_d.is_cuda):
            raise _lib.EnvidrError("render: rays must live on the GPU")
        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        N = rays_o.shape[0]
        dev = rays_o.device
        res = out if out is not None else {}

        def buf(name, *shape):
            if name not in res:
                res[name] = torch.empty(*shape, dtype=torch.float32, device=dev)
            return res[name]

        o = RenderOut()
        o.image = buf("image", N, 3).data_ptr()
        o.depth = buf("depth", N).data_ptr()
        o.weights_sum = buf("weights_sum", N).data_ptr()
        if extras:
            o.normal_image = buf("normal_image", N, 3).data_ptr()
            o.diffuse_image = buf("diffuse_image", N, 3).data_ptr()
            o.specular_image = buf("specular_image", N, 3).data_ptr()
            o.roughness_image = buf("roughness_image", N).data_ptr()
        if stats:
            res["stats"] = torch.zeros(12, dtype=torch.int64, device=dev)
            o.stats = res["stats"].data_ptr()
        self.set_env_rotation(env_rot_radian)
        self.desc.geometry_only = int(bool(geometry_only))
        self.desc.r_images = None
        if r_images is not None and not geometry_only:
            if not self.desc.renv_blob:
                raise _lib.EnvidrError("render: r_images given but the model has no renv MLP")
            r_images = r_images.contiguous().view(-1, 4).float()
            if r_images.shape[0] != N or not r_images.is_cuda:
                raise _lib.EnvidrError("render: r_images must be [N,4] on the GPU")
            self.desc.r_images = r_images.data_ptr()
        self.desc.ray_cost = None
        if ray_cost is not None:
            if ray_cost.dtype != torch.int16 or ray_cost.numel() != N or not ray_cost.is_cuda or not ray_cost.is_contiguous():
                raise _lib.EnvidrError("render: ray_cost must be a contiguous int16 [N] tensor on the GPU")
            self.desc.ray_cost = ray_cost.data_ptr()
        # work-list scratch from torch's stream-aware allocator: renders issued on different streams never share it
        need = int(self.lib.envidr_render_scratch_bytes(N))
        scratch = torch.empty((need + 3) // 4, dtype=torch.int32, device=dev)
        self.desc.scratch, self.desc.scratch_bytes = scratch.data_ptr(), scratch.numel() * 4
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = self.lib.envidr_render_rays(ctypes.byref(self.desc), rays_o.data_ptr(), rays_d.data_ptr(), N, ctypes.byref(o),
                                         self.counter.data_ptr(), stream)
        if rc:
            raise _lib.EnvidrError(f"envidr_render_rays failed ({rc}): {self.lib.envidr_last_error().decode()}")
        return res
