"""CPU-side checks of the C-ABI boundary: the shared library loads without a GPU, exports every entry
point the headers declare, the host-side weight packing matches its specification, and the product
package never touches the oracle."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _declared(header: str) -> set[str]:
    text = (ROOT / "include" / header).read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(envidr_[a-zA-Z0-9_]+)\s*\(", text))


def test_library_exports_every_declared_symbol():
    from envidr_amd import _lib
    lib = _lib.load()
    declared = _declared("envidr_amd.h") | _declared("envidr_render.h")
    assert len(declared) >= 28
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    # and the Python binding table covers every operator of envidr_amd.h
    ops = {"envidr_" + n for n in _lib.SIGNATURES}
    assert ops <= declared
    assert lib.envidr_abi_version() == 10


def test_signature_table_matches_header_arity():
    from envidr_amd import _lib
    text = (ROOT / "include" / "envidr_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, sig in _lib.SIGNATURES.items():
        m = re.search(r"\benvidr_" + name + r"\s*\((.*?)\)\s*;", text, flags=re.S)
        assert m, name
        params = [p.strip() for p in m.group(1).split(",")]
        assert params[-1].startswith("envidr_stream_t"), name
        assert len(params) - 1 == len(sig), f"{name}: header has {len(params) - 1} args, table {len(sig)}"
        for p, k in zip(params, sig):
            is_ptr = "*" in p
            assert is_ptr == (k == "p"), f"{name}: '{p}' vs kind {k}"
            if not is_ptr:
                want = {"uint32_t": "u", "float": "f", "int": "i"}[p.split()[0]]
                assert want == k, f"{name}: '{p}' vs kind {k}"


def test_errors_without_gpu_are_reported():
    """argument validation happens before any launch, so it is testable on the CPU"""
    from envidr_amd import _lib
    lib = _lib.load()
    lib.envidr_sh_encode_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                             ctypes.c_void_p, ctypes.c_void_p]
    rc = lib.envidr_sh_encode_forward(None, None, 4, 3, 9, None, None)
    assert rc == -1 and b"degree" in lib.envidr_last_error()
    rc = lib.envidr_sh_encode_forward(None, None, 0, 3, 4, None, None)      # empty batch: successful no-op
    assert rc == 0


def test_render_entry_points_validate_their_descriptor():
    """envidr_render_rays / envidr_shade_samples reject inconsistent descriptors before launching anything"""
    from envidr_amd import _lib, fused
    lib = _lib.load()
    fused._bind_render(lib)
    d = fused.RenderDesc()
    o = fused.RenderOut()
    assert lib.envidr_render_rays(ctypes.byref(d), None, None, 0, ctypes.byref(o), None, None) == 0      # N = 0: no-op
    assert lib.envidr_render_rays(None, None, None, 8, ctypes.byref(o), None, None) == -1
    assert b"null descriptor" in lib.envidr_last_error()
    assert lib.envidr_render_rays(ctypes.byref(d), None, None, 8, ctypes.byref(o), None, None) == -1
    assert b"null ray pointers" in lib.envidr_last_error()
    assert lib.envidr_shade_samples(ctypes.byref(d), None, None, None, 0, None, 0, 0, None, None, None) == 0   # M = 0
    assert lib.envidr_shade_samples(ctypes.byref(d), None, None, None, 0, None, 0, 4, None, None, None) == -1
    assert b"null pointer" in lib.envidr_last_error()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.envidr_shade_samples(ctypes.byref(d), p, p, p, 12, p, 1, 4, p, p, None) == -1
    assert b"null weight blob" in lib.envidr_last_error()
    d.env_blob = d.head_blob = p
    assert lib.envidr_shade_samples(ctypes.byref(d), p, p, p, 7, p, 1, 4, p, p, None) == -1
    assert b"geo_feat_stride" in lib.envidr_last_error()
    d.dir_sh_degree = 3
    assert lib.envidr_shade_samples(ctypes.byref(d), p, p, p, 12, p, 1, 4, p, p, None) == -1
    assert b"unsupported dir_sh_degree" in lib.envidr_last_error()
    # geometry pipeline entry points
    d = fused.RenderDesc()
    ex = fused.GeometryExport()
    assert lib.envidr_geometry_pass(ctypes.byref(d), None, None, 0, ctypes.byref(o), ctypes.byref(ex), None, 0, 0, None) == 0      # N = 0
    assert lib.envidr_geometry_pass(None, None, None, 8, ctypes.byref(o), ctypes.byref(ex), None, 0, 0, None) == -1
    assert b"null descriptor" in lib.envidr_last_error()
    assert lib.envidr_geometry_pass(ctypes.byref(d), p, p, 8, ctypes.byref(o), ctypes.byref(ex), None, 0, 0, None) == -1
    assert b"workspace" in lib.envidr_last_error()
    assert lib.envidr_geometry_workspace_bytes(640000, 16_000_000) > 16_000_000 * 92
    so = fused.SamplesOut()
    assert lib.envidr_geometry_eval(ctypes.byref(d), None, None, 0, None, ctypes.byref(so), None) == 0                              # M = 0
    assert lib.envidr_geometry_eval(ctypes.byref(d), None, None, 4, None, ctypes.byref(so), None) == -1
    assert b"null sample pointers" in lib.envidr_last_error()


def _tile_row(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


@pytest.mark.parametrize("m_out,k_in,order,transpose", [(64, 32, 0, False), (256, 72, 0, False), (256, 256, 1, False),
                                                        (12, 256, 1, False), (15, 64, 1, False), (64, 64, 1, True),
                                                        (64, 32, 1, True), (3, 32, 1, False), (160, 38, 0, False)])
def test_pack_linear_layout(m_out, k_in, order, transpose):
    """dst[(step * tiles + tile) * 64 + half * 32 + i] = W'[32 tile + i][k(step, half)] with zero padding"""
    from envidr_amd.fused import pack_linear
    rng = np.random.default_rng(0)
    W = rng.normal(size=(m_out, k_in)).astype(np.float32)
    got = pack_linear(W, order, transpose)
    Wl = W.T if transpose else W                      # logical layer K -> M
    M, K = Wl.shape
    steps = (K + 1) // 2 if order == 0 else ((K + 31) // 32 * 32) // 2
    tiles = (M + 31) // 32
    assert got.shape == (steps * tiles * 64,)
    for s in range(steps):
        for h in range(2):
            k = 2 * s + h if order == 0 else 32 * (s >> 4) + _tile_row(s & 15, h)
            for t in range(tiles):
                seg = got[(s * tiles + t) * 64 + h * 32:(s * tiles + t) * 64 + h * 32 + 32]
                want = np.zeros(32, np.float32)
                rows = np.arange(32 * t, min(32 * t + 32, M))
                if k < K:
                    want[: rows.size] = Wl[rows, k]
                assert np.array_equal(seg, want)


def test_pack_rowvec_layout():
    from envidr_amd.fused import pack_rowvec
    v = np.arange(40, dtype=np.float32) + 1
    got = pack_rowvec(v)
    assert got.shape == (64,)
    for t in range(2):
        for h in range(2):
            for r in range(16):
                m = 32 * t + _tile_row(r, h)
                assert got[t * 32 + h * 16 + r] == (v[m] if m < 40 else 0)


def test_product_never_imports_the_oracle():
    """the shipped package must not route through oracle/ (or any CPU fallback)"""
    for py in (ROOT / "envidr_amd").rglob("*.py"):
        text = py.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{py} imports the oracle"
    for src in (ROOT / "envidr_amd" / "csrc").glob("*.h*"):
        text = src.read_text()
        assert not re.search(r'#include\s*[<"][^>"]*oracle', text), f"{src} includes oracle code"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from envidr_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_lib.EnvidrError):
        _lib.load()


def test_every_operator_takes_an_empty_batch_as_a_no_op():
    """a zero element count returns ENVIDR_OK before anything is dereferenced or launched (the reference's wrappers can hand
    their kernels empty tensors: an image whose rays all miss the box has n_alive == 0) -- checkable without a GPU: every
    pointer is NULL here"""
    from envidr_amd import _lib
    P = None
    empty = {
        "near_far_from_aabb": (P, P, P, 0, 0.05, P, P),
        "get_rays": (P, 1.0, 1.0, 0.5, 0.5, 0, 0, 0, P, P),
        "sph_from_ray": (P, P, 1.0, 0, P),
        "morton3D": (P, 0, P), "morton3D_invert": (P, 0, P),
        "packbits": (P, 0, 0.01, P),
        "get_scatter_idx": (P, 0, P),
        "march_rays_train": (P, P, P, 1.0, 0.0, 1024, 0, 0, 1, 128, 0, P, P, P, P, P, P, P, P),
        "composite_rays_train_forward": (P, P, P, P, 0, 0, 1e-4, 1, 0, P, P, P, P),
        "composite_rays_train_backward": (P, P, P, P, P, P, P, P, P, P, 0, 0, 1e-4, P, P, 1, 0),
        "march_rays": (0, 4, P, P, P, P, 1.0, 0.0, 1024, 1, 128, P, P, P, P, P, P, P),
        "composite_rays": (0, 4, 1e-4, 1, 0, P, P, P, P, P, P, P, P),
        "hash_encode_forward": (P, P, P, P, 0, 3, 2, 16, 0.4, 16, 1, P),
        "hash_encode_backward": (P, P, P, P, P, 0, 3, 2, 16, 0.4, 16, 1, P, P),
        "hash_encode_second_backward": (P, P, P, P, 0, 3, 2, 16, 0.4, 16, 1, P, P, P, P),
        "grid_encode_forward": (P, P, P, P, 0, 3, 2, 8, 0.4, 16, P, 0, 0),
        "grid_encode_backward": (P, P, P, P, P, 0, 3, 2, 8, 0.4, 16, P, P, 0, 0),
        "hash_encode_forward_f16": (P, P, P, P, 0, 3, 2, 16, 0.4, 16, 1, P),
        "hash_encode_backward_f16": (P, P, P, P, P, 0, 3, 2, 16, 0.4, 16, 1, P, P),
        "hash_encode_second_backward_f16": (P, P, P, P, 0, 3, 2, 16, 0.4, 16, 1, P, P, P, P),
        "grid_encode_forward_f16": (P, P, P, P, 0, 3, 2, 8, 0.4, 16, P, 0, 0),
        "grid_encode_backward_f16": (P, P, P, P, P, 0, 3, 2, 8, 0.4, 16, P, P, 0, 0),
        "freq_encode_forward": (P, 0, 3, 4, 27, P),
        "freq_encode_backward": (P, P, 0, 3, 4, 27, P),
        "sh_encode_forward": (P, P, 0, 3, 4, P),
        "sh_encode_backward": (P, P, 0, 3, 4, P, P),
        "sh_encode_forward_f16": (P, P, 0, 3, 4, P),
        "sh_encode_backward_f16": (P, P, 0, 3, 4, P, P),
        "ide_encode_forward": (P, P, 0.5, 0, 5, P),
        "ide_encode_backward": (P, P, P, 0.5, 0, 5, P, P),
    }
    # (compact_alive is the exception: with nothing alive it still writes *out_count = 0 -- tests/test_ops_gpu.py)
    assert set(empty) | {"compact_alive"} == set(_lib.SIGNATURES), set(_lib.SIGNATURES) ^ set(empty)
    for name, args in empty.items():
        _lib.call(name, *args, stream=0)          # raises EnvidrError on any non-zero return code


def test_round5_entry_points_validate_their_arguments():
    """the env-sphere operators and the weight gradient reject null pointers / impossible sizes before launching anything (no GPU needed)"""
    from envidr_amd import _lib, fused
    lib = _lib.load()
    fused._bind_render(lib)
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.envidr_sphere_intersections(None, None, 0, 1.0, None, None, None, None) == 0                      # N = 0: no-op
    assert lib.envidr_sphere_intersections(None, None, 4, 1.0, None, None, None, None) == -1 and b"null pointer" in lib.envidr_last_error()
    assert lib.envidr_shell_samples(None, None, None, None, None, None, 0.002, 0, 12, None, None, None, None) == 0   # M = 0
    assert lib.envidr_shell_samples(None, None, None, None, None, None, 0.002, 4, 12, None, None, None, None) == -1
    args = [None] * 10 + [0, 0, 12, 0.002, 1.0] + [None] * 8
    assert lib.envidr_composite_shell(*args) == 0                                                                 # N = 0
    args[10] = 4
    assert lib.envidr_composite_shell(*args) == -1 and b"null pointer" in lib.envidr_last_error()
    # S = 0 and images without their per-sample inputs
    ok = [p, p, p, p, None, None, p, p, p, p, 4, 2, 0, 0.002, 1.0, p, p, p, None, None, None, None, None]
    assert lib.envidr_composite_shell(*ok) == -1 and b"S must be" in lib.envidr_last_error()
    ok[12] = 12
    ok[18] = p                                                                                                    # normal_image without normals
    assert lib.envidr_composite_shell(*ok) == -1 and b"normal_image" in lib.envidr_last_error()
    assert lib.envidr_linear_weight_grad_workspace_bytes(0, 8, 8) == 0 and lib.envidr_linear_weight_grad_workspace_bytes(1000, 256, 256) > 0
    assert lib.envidr_linear_weight_grad(None, None, 10, 0, 4, p, None, 0, None, 0, None) == -1 and b"empty layer" in lib.envidr_last_error()
    assert lib.envidr_linear_weight_grad(None, None, 10, 4, 4, p, None, 0, None, 0, None) == -1 and b"null pointer" in lib.envidr_last_error()
    # ABI 9: envidr_linear_rows(x, ldx, M, K, W, w_stride_out, w_stride_in, N, bias, act, ldact, epilogue, y, ldy, stream)
    assert lib.envidr_linear_rows(p, 8, 10, 6, p, 6, 1, 4, None, None, 0, 0, p, 4, None) == -1 and b"multiple of 4" in lib.envidr_last_error()
    assert lib.envidr_linear_rows(p, 8, 10, 8, p, 8, 1, 4, None, None, 0, 7, p, 4, None) == -1 and b"unknown epilogue" in lib.envidr_last_error()
    assert lib.envidr_linear_rows(p, 8, 0, 8, p, 8, 1, 4, None, None, 0, 0, p, 4, None) == 0                       # an empty batch is not an error
    assert lib.envidr_linear_rows(None, 8, 10, 8, p, 8, 1, 4, None, None, 0, 0, p, 4, None) == -1 and b"null pointer" in lib.envidr_last_error()
    assert lib.envidr_linear_rows(p, 6, 10, 8, p, 8, 1, 4, None, None, 0, 0, p, 4, None) == -1 and b"16-byte aligned" in lib.envidr_last_error()
    assert lib.envidr_linear_rows(p, 8, 10, 8, p, 8, 1, 4, None, None, 0, 2, p, 4, None) == -1 and b"needs a bias" in lib.envidr_last_error()
    assert lib.envidr_linear_rows(p, 8, 10, 8, p, 8, 1, 4, None, None, 0, 3, p, 4, None) == -1 and b"needs act" in lib.envidr_last_error()
    assert lib.envidr_linear_rows(p, 8, 10, 8, p, 8, 1, 4, None, None, 0, 0, p, 3, None) == -1 and b"ldy" in lib.envidr_last_error()
