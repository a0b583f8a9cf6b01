"""BASELINE sizes (800 x 800 = 640 000 rays, 1024 max samples per ray, full hash table): the oracle's Python render
loop is too slow for a whole frame, so the fused path is checked through size-independent properties, and the
bandwidth-bound operators bit-for-bit against the C oracle (which is fast enough at this size)."""
import numpy as np
import pytest

from envidr_amd import parallel, scenes
from tests.util import bits_equal, rel_l2, run_op

pytestmark = pytest.mark.gpu
H = W = 800
N = H * W


@pytest.fixture(scope="module")
def frame():
    import torch
    from envidr_amd.fused import FusedRenderer
    scene = scenes.toaster_scene()
    r = FusedRenderer.from_scene(scene)
    ro, rd = (torch.from_numpy(a).cuda() for a in scenes.camera_rays(H, W))
    res = {k: v.clone() for k, v in r.render(ro, rd, 0.9, extras=True, stats=True).items()}
    torch.cuda.synchronize()
    return scene, r, ro, rd, res


def test_full_frame_invariants(frame):
    import torch
    _, _, _, _, res = frame
    assert int(res["stats"][2]) == N                                   # every ray handled exactly once
    assert 5_000_000 < int(res["stats"][0]) < 12_000_000                # samples shaded (about 12 per ray)
    for k in ("image", "depth", "weights_sum", "normal_image", "diffuse_image", "specular_image", "roughness_image"):
        assert torch.isfinite(res[k]).all(), k
    ws = res["weights_sum"]
    assert float(ws.min()) >= 0 and float(ws.max()) <= 1 + 1e-6
    miss = ws == 0
    assert 0.3 < float(miss.float().mean()) < 0.8
    assert torch.all(res["image"][miss] == 1.0) and torch.all(res["depth"][miss] == 0)      # background only
    hit = ~miss
    nn = res["normal_image"][hit].norm(dim=-1)
    assert torch.allclose(nn, torch.ones_like(nn), atol=1e-5)          # composited normals are re-normalised
    assert float(res["image"][hit].min()) >= 0 and float(res["image"][hit].max()) <= 2.0 + 1e-5   # sigmoid + sigmoid + bg share


def test_result_of_a_ray_does_not_depend_on_its_batch(frame):
    """permutation, subset and tile-sharded renders reproduce the full frame bit for bit (this is also the
    single-frame multi-GPU decomposition: 8 interleaved tile shards)"""
    import torch
    _, r, ro, rd, res = frame
    keys = ("image", "depth", "weights_sum", "normal_image", "diffuse_image", "specular_image", "roughness_image")
    g = torch.Generator().manual_seed(3)
    perm = torch.randperm(N, generator=g).cuda()
    out = r.render(ro[perm], rd[perm], 0.9, extras=True)
    torch.cuda.synchronize()
    for k in keys:
        assert torch.equal(out[k], res[k][perm]), f"permutation: {k}"
    sub = perm[:5003]
    out = r.render(ro[sub], rd[sub], 0.9, extras=True)
    torch.cuda.synchronize()
    for k in keys:
        assert torch.equal(out[k], res[k][sub]), f"subset: {k}"
    parts = []
    for rank in range(8):
        idx = parallel.tile_shard(H, W, rank, 8).cuda()
        parts.append(r.render(ro[idx], rd[idx], 0.9, extras=False)["image"].clone())
    torch.cuda.synchronize()
    assert torch.equal(parallel.assemble_frame(parts, H, W), res["image"])


def test_fused_equals_operator_loop_on_a_full_size_sample(frame):
    """5 000 rays drawn from the 800 x 800 frame through the reference-shaped operator loop (HIP operators + rocBLAS
    GEMMs, the reference's n_step schedule) vs the same rays of the fused full frame"""
    import torch
    from envidr_amd.nerf.network import NeRFNetwork
    scene, _, ro, rd, res = frame
    model = NeRFNetwork.from_scene(scene)
    g = torch.Generator().manual_seed(11)
    idx = torch.randperm(N, generator=g)[:5000].cuda()
    opt = model.opt
    out = model.render(ro[idx][None], rd[idx][None], staged=True, bg_color=1, perturb=False, get_normal_image=False,
                       env_rot_radian=0.9, fused=False, max_steps=opt.max_steps, T_thresh=opt.T_thresh, dt_gamma=opt.dt_gamma)
    torch.cuda.synchronize()
    for k in ("image", "depth", "weights_sum", "diffuse_image", "specular_image"):
        a = out[k].reshape(5000, -1).cpu().numpy()
        b = res[k][idx].reshape(5000, -1).cpu().numpy()
        assert rel_l2(a, b) <= 2e-5, f"{k}: {rel_l2(a, b):.3e}"


def test_env_rotation_by_a_full_turn_is_the_identity(frame):
    """rot_theta(2 pi) differs from the identity by 2.4e-7 in its off-diagonal entries: the frames agree to fp32 noise"""
    import math
    import torch
    _, r, ro, rd, _ = frame
    a = r.render(ro[:100_000], rd[:100_000], None, extras=False)["image"].clone()
    b = r.render(ro[:100_000], rd[:100_000], 2 * math.pi, extras=False)["image"]
    torch.cuda.synchronize()
    assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) <= 1e-5


def test_operators_bit_exact_at_full_size():
    """near/far, the inference marcher's first call and the hash-grid lookup (+ dy_dx) at 640 000 rays / points on the
    full 6.1 M-row table, bit for bit against the C oracle; morton / packbits over the full 128^3 grid"""
    scene = scenes.toaster_scene()
    ro, rd = scenes.camera_rays(H, W)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    args = ("near_far_from_aabb", ro, rd, aabb, N, 0.2, np.zeros(N, np.float32), np.zeros(N, np.float32))
    got, want = run_op("hip", *args), run_op("oracle", *args)
    nears, fars = want[-2], want[-1]
    assert bits_equal(got[-2], nears) and bits_equal(got[-1], fars)
    M = N + 128 - N % 128
    alive = np.arange(N, dtype=np.int32)
    args = ("march_rays", N, 1, alive, nears.copy(), ro, rd, 1.0, 0.0, 1024, 1, 128, scene.bitfield, nears, fars,
            np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32), np.zeros(N, np.float32))
    got, want = run_op("hip", *args), run_op("oracle", *args)
    for i in (-4, -3, -2):                                           # xyzs, dirs, deltas
        assert bits_equal(got[i], want[i])
    xyzs = want[-4][:N]
    assert int((want[-2][:, 0] > 0).sum()) > 200_000                 # first samples of all hit rays
    inputs = ((xyzs + 1) / 2).astype(np.float32)
    L, C = 16, 2
    out, dydx = np.zeros((L, N, C), np.float32), np.zeros((N, L * 3 * C), np.float32)
    args = ("hash_encode_forward", inputs, scene.table, scene.offsets.astype(np.int32), out, N, 3, C, L,
            float(np.log2(scene.per_level_scale)), 16, 1, dydx)
    got, want = run_op("hip", *args), run_op("oracle", *args)
    assert bits_equal(got[3], want[3]) and bits_equal(got[4], want[4])
    cells = 128 ** 3
    coords = np.stack(np.unravel_index(np.arange(cells), (128, 128, 128)), -1).astype(np.int32)
    args = ("morton3D", coords, cells, np.zeros(cells, np.int32))
    got, want = run_op("hip", *args), run_op("oracle", *args)
    assert np.array_equal(got[-1], want[-1]) and len(np.unique(got[-1])) == cells
    args = ("morton3D_invert", want[-1], cells, np.zeros((cells, 3), np.int32))
    assert np.array_equal(run_op("hip", *args)[-1], coords)
    grid = np.random.default_rng(0).uniform(-1, 30, size=cells).astype(np.float32)
    args = ("packbits", grid, cells // 8, 10.0, np.zeros(cells // 8, np.uint8))
    assert np.array_equal(run_op("hip", *args)[-1], run_op("oracle", *args)[-1])


# ---- BASELINE configs[1] and [3] at 800 x 800: the same size-independent properties as for the toaster network ----------------
KEYS7 = ("image", "depth", "weights_sum", "normal_image", "diffuse_image", "specular_image", "roughness_image")


def _shard_reassembly(render, ro, rd, full_image):
    """the frame as 8 interleaved tile shards (the strong-scaling decomposition of parallel.tile_shard) == the whole frame"""
    import torch
    parts = [render(ro[idx], rd[idx])["image"].clone() for idx in (parallel.tile_shard(H, W, rank, 8).cuda() for rank in range(8))]
    torch.cuda.synchronize()
    assert torch.equal(parallel.assemble_frame(parts, H, W), full_image)


def test_configs1_no_env_network_at_full_size():
    """configs[1] (hash-grid SDF + diffuse / specular heads on SH encodings, no environment MLP) through the frame pipeline bench.py
    runs: invariants, permutation / subset / tile-shard independence bit for bit, and 5 000 rays against the operator loop"""
    import torch
    from envidr_amd.fused import FusedOptions, FusedRenderer
    from envidr_amd.nerf.network import NeRFNetwork
    scene = scenes.lego_scene()
    r = FusedRenderer.from_scene(scene, FusedOptions(dir_sh_degree=4))
    ro, rd = (torch.from_numpy(a).cuda() for a in scenes.camera_rays(H, W))
    res = {k: v.clone() for k, v in r.render_frame(ro, rd, None, out={}, image_width=W).items() if torch.is_tensor(v)}
    for k in KEYS7:
        assert torch.isfinite(res[k]).all(), k
    ws = res["weights_sum"]
    miss = ws == 0
    assert 0.3 < float(miss.float().mean()) < 0.8 and torch.all(res["image"][miss] == 1.0)
    assert float(ws.max()) <= 1 + 1e-6 and 3_000_000 < int(r._frame["last"][1]) < 20_000_000
    g = torch.Generator().manual_seed(4)
    perm = torch.randperm(N, generator=g).cuda()
    out = r.render_frame(ro[perm], rd[perm], None, out={})
    for k in KEYS7:
        assert torch.equal(out[k], res[k][perm]), f"permutation: {k}"
    sub = perm[:5003]
    out = r.render_frame(ro[sub], rd[sub], None, out={})
    for k in KEYS7:
        assert torch.equal(out[k], res[k][sub]), f"subset: {k}"
    _shard_reassembly(lambda o, d: r.render_frame(o, d, None, out={}), ro, rd, res["image"])
    # 5 000 rays through the reference-shaped operator loop
    from envidr_amd.nerf.options import toaster_options
    from tests.test_dropin_gpu import LEGO
    opt = toaster_options(**LEGO)
    model = NeRFNetwork.from_scene(scene, opt)
    idx = perm[:5000]
    loop = model.render(ro[idx][None], rd[idx][None], staged=True, bg_color=1, perturb=False, get_normal_image=False, env_rot_radian=None,
                        fused=False, max_steps=opt.max_steps, T_thresh=opt.T_thresh, dt_gamma=opt.dt_gamma)
    torch.cuda.synchronize()
    for k in ("image", "depth", "weights_sum", "diffuse_image", "specular_image"):
        a, b = loop[k].reshape(5000, -1).cpu().numpy(), res[k][idx].reshape(5000, -1).cpu().numpy()
        assert rel_l2(a, b) <= 2e-5, f"{k}: {rel_l2(a, b):.3e}"


def test_configs3_indirect_three_pass_at_full_size():
    """configs[3] (use_renv + indir_ref on the concave torus scene) through NeRFRenderer.render: the masked three-pass form at
    800 x 800 -- invariants, tile-shard reassembly bit for bit, and 5 000 rays against the reference-shaped form (boolean-mask
    gathers + operator loop)"""
    import torch
    from envidr_amd.nerf.network import NeRFNetwork
    from envidr_amd.nerf.options import toaster_options
    opt = toaster_options(indir_ref=True)
    model = NeRFNetwork.from_scene(scenes.toaster_scene(shape=scenes.torus(), seed=3), opt)
    ro, rd = (torch.from_numpy(a).cuda() for a in scenes.camera_rays(H, W))
    kw = dict(staged=True, bg_color=1, perturb=False, get_normal_image=True, env_rot_radian=0.4, max_steps=opt.max_steps, T_thresh=opt.T_thresh,
              dt_gamma=opt.dt_gamma, early_stop_steps=-1)

    def render(o, d, **more):
        out = model.render(o[None], d[None], fused=True, **kw, **more)
        return {k: v.reshape(o.shape[0], -1).squeeze(-1).clone() for k, v in out.items() if torch.is_tensor(v)}
    res = render(ro, rd, image_width=W)
    torch.cuda.synchronize()
    for k in KEYS7:
        assert torch.isfinite(res[k]).all(), k
    ws = res["weights_sum"]
    assert 0.2 < float((ws > 0.3).float().mean()) < 0.9 and float(ws.max()) <= 1 + 1e-6
    assert torch.all(res["image"][ws == 0] == 1.0)
    _shard_reassembly(lambda o, d: render(o, d), ro, rd, res["image"].reshape(N, 3))
    g = torch.Generator().manual_seed(6)
    idx = torch.randperm(N, generator=g)[:5000].cuda()
    sub = render(ro[idx], rd[idx])
    for k in KEYS7:
        assert torch.equal(sub[k], res[k][idx]), f"subset: {k}"
    loop = model.render(ro[idx][None], rd[idx][None], fused=False, **kw)
    torch.cuda.synchronize()
    for k in ("image", "depth", "weights_sum", "diffuse_image", "specular_image"):
        a, b = loop[k].reshape(5000, -1).cpu().numpy(), res[k][idx].reshape(5000, -1).cpu().numpy()
        assert rel_l2(a, b) <= 5e-5, f"{k}: {rel_l2(a, b):.3e}"
