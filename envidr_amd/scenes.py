"""Synthetic scenes for tests and benchmarks (no datasets or checkpoints exist offline).

Everything here is deterministic numpy on the host: cameras, rays, an analytic occupancy bitfield
and seeded weights.  The conventions restate the reference's so that its own ray generator and
camera loader would produce the same numbers:
  * camera   -- `pose_spherical` (nerf/sph_loader.py:67-76) followed by `nerf_matrix_to_ngp`
                (nerf/provider.py:32-40); focal = W / (2 tan(camera_angle_x / 2)),
                principal point at the image centre (sph_loader.py:99-108);
  * rays     -- pixel centres at +0.5, unit directions, rotated by the pose (nerf/utils.py:193-207);
  * bitfield -- Morton-ordered, bit i of byte n = voxel 8n+i (raymarching.cu:267-289, renderer.py:110);
  * hash table layout / level sizes -- hashencoder/hashgrid.py:115-152.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

CAMERA_ANGLE_X = 0.6194058656692505  # configs/env_dataset_config.ini


# ------------------------------------------------------------------------------------------------
# cameras and rays
# ------------------------------------------------------------------------------------------------
def pose_spherical(theta_deg: float, phi_deg: float, radius: float) -> np.ndarray:
    """camera-to-world of an orbit camera looking at the origin (blender convention)."""
    th, ph = math.radians(theta_deg), math.radians(phi_deg)
    trans = np.eye(4)
    trans[2, 3] = radius
    rot_phi = np.array([[1, 0, 0, 0], [0, math.cos(ph), -math.sin(ph), 0], [0, math.sin(ph), math.cos(ph), 0], [0, 0, 0, 1]])
    rot_th = np.array([[math.cos(th), 0, -math.sin(th), 0], [0, 1, 0, 0], [math.sin(th), 0, math.cos(th), 0], [0, 0, 0, 1]])
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)
    return flip @ rot_th @ rot_phi @ trans


def nerf_matrix_to_ngp(pose: np.ndarray, scale: float = 0.33, offset=(0.0, 0.0, 0.0)) -> np.ndarray:
    """axis permutation (x,y,z)->(y,z,x) with flipped y/z camera axes, translation scaled."""
    out = np.eye(4, dtype=np.float32)
    for row, src in enumerate((1, 2, 0)):
        out[row, 0] = pose[src, 0]
        out[row, 1] = -pose[src, 1]
        out[row, 2] = -pose[src, 2]
        out[row, 3] = pose[src, 3] * scale + offset[row]
    return out


def intrinsics_for(H: int, W: int, camera_angle_x: float = CAMERA_ANGLE_X) -> np.ndarray:
    focal = W / (2 * math.tan(camera_angle_x / 2))
    return np.array([focal, focal, W / 2, H / 2], dtype=np.float64)


def get_rays(pose: np.ndarray, intrinsics, H: int, W: int) -> tuple[np.ndarray, np.ndarray]:
    """full-image rays, row-major pixels; fp32 arithmetic in the reference's order."""
    fx, fy, cx, cy = (np.float32(v) for v in intrinsics)
    i = (np.arange(W, dtype=np.float32)[None, :] + np.float32(0.5)).repeat(H, 0).reshape(-1)
    j = (np.arange(H, dtype=np.float32)[:, None] + np.float32(0.5)).repeat(W, 1).reshape(-1)
    xs = (i - cx) / fx
    ys = (j - cy) / fy
    d = np.stack([xs, ys, np.ones_like(xs)], -1).astype(np.float32)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True).astype(np.float32)
    R = pose[:3, :3].astype(np.float32)
    rays_d = (d @ R.T).astype(np.float32)
    rays_o = np.broadcast_to(pose[:3, 3].astype(np.float32), rays_d.shape).copy()
    return np.ascontiguousarray(rays_o), np.ascontiguousarray(rays_d)


def camera_rays(H: int, W: int, theta: float = 30.0, phi: float = -20.0, radius: float = 4.0, scale: float = 0.65):
    pose = nerf_matrix_to_ngp(pose_spherical(theta, phi, radius), scale=scale)
    return get_rays(pose, intrinsics_for(H, W), H, W)


def rot_theta3(theta: float) -> np.ndarray:
    """3x3 block of the reference's rot_theta (nerf/utils.py:48-52), used for env rotation."""
    c, s = math.cos(theta), math.sin(theta)
    return np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]], dtype=np.float64)


# ------------------------------------------------------------------------------------------------
# occupancy bitfield
# ------------------------------------------------------------------------------------------------
def _part1by2(v: np.ndarray) -> np.ndarray:
    v = v.astype(np.uint32)
    v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
    v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
    v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
    v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
    return v


def morton3d(x, y, z) -> np.ndarray:
    return _part1by2(x) | (_part1by2(y) << np.uint32(1)) | (_part1by2(z) << np.uint32(2))


def occupancy_bitfield(inside_fn, bound: float = 1.0, H: int = 128, cascades: int = 1) -> np.ndarray:
    """bitfield[C*H^3/8] with voxel (ix,iy,iz) of cascade c set when inside_fn(centres) is True."""
    out = np.zeros(cascades * H ** 3 // 8, dtype=np.uint8)
    ax = np.arange(H)
    ix, iy, iz = np.meshgrid(ax, ax, ax, indexing="ij")
    code = morton3d(ix.ravel(), iy.ravel(), iz.ravel()).astype(np.int64)
    for c in range(cascades):
        half = min(2.0 ** c, bound)
        ctr = lambda a: ((a.ravel() + 0.5) / H * 2 - 1) * half
        occ = inside_fn(np.stack([ctr(ix), ctr(iy), ctr(iz)], -1))
        idx = c * H ** 3 + code[occ]
        np.bitwise_or.at(out, idx >> 3, (1 << (idx & 7)).astype(np.uint8))
    return out


def shell(radius: float = 0.5, half_thickness: float = 0.05):
    return lambda p: np.abs(np.linalg.norm(p, axis=-1) - radius) < half_thickness


def ball(radius: float = 0.6):
    return lambda p: np.linalg.norm(p, axis=-1) < radius


def torus(R: float = 0.55, r: float = 0.2, pad: float = 0.03):
    def f(p):
        q = np.sqrt(p[:, 0] ** 2 + p[:, 2] ** 2) - R
        return np.sqrt(q ** 2 + p[:, 1] ** 2) < r + pad
    return f


# ------------------------------------------------------------------------------------------------
# hash-grid geometry and seeded parameters
# ------------------------------------------------------------------------------------------------
def hash_level_offsets(input_dim=3, num_levels=16, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048):
    """returns (offsets int32[L+1], per_level_scale) exactly as HashEncoder.__init__ sizes them."""
    per_level_scale = float(np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1)))
    offs, off = [], 0
    for i in range(num_levels):
        res = int(np.ceil(base_resolution * per_level_scale ** i))
        offs.append(off)
        off += min(2 ** log2_hashmap_size, res ** input_dim)
    offs.append(off)
    return np.array(offs, dtype=np.int32), per_level_scale


def grid_level_offsets(input_dim=3, num_levels=16, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048,
                       align_corners=False):
    """GridEncoder sizing (gridencoder/grid.py:108-121): (res [+1])^D rows rounded up to x8."""
    per_level_scale = float(np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1)))
    offs, off = [], 0
    for i in range(num_levels):
        res = int(np.ceil(base_resolution * per_level_scale ** i))
        n = min(2 ** log2_hashmap_size, (res if align_corners else res + 1) ** input_dim)
        n = int(np.ceil(n / 8) * 8)
        offs.append(off)
        off += n
    offs.append(off)
    return np.array(offs, dtype=np.int32), per_level_scale


def xavier_linear(rng: np.random.Generator, fan_in: int, fan_out: int, gain: float = math.sqrt(2.0)):
    """xavier-uniform weight [out,in] with relu gain and a zero bias (net_init.py style init)."""
    a = gain * math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-a, a, size=(fan_out, fan_in)).astype(np.float32), np.zeros(fan_out, np.float32)


@dataclass
class SceneParams:
    """All state the render path consumes for one synthetic scene (host numpy arrays)."""
    bitfield: np.ndarray
    offsets: np.ndarray
    per_level_scale: float
    table: np.ndarray                      # [rows, 2]
    mlps: dict = field(default_factory=dict)   # name -> list of (W[out,in], b[out])
    beta: float = 0.01
    bound: float = 1.0
    grid_size: int = 128
    cascades: int = 1


def make_mlp(rng, dims):
    return [xavier_linear(rng, dims[i], dims[i + 1]) for i in range(len(dims) - 1)]


def toaster_scene(table_scale: float = 0.1, shape=None, sdf_bias: float = 0.005, beta: float = 0.01,
                  hidden_env: int = 256, ide_deg: int = 5, seed: int = 0, arrays: bool = True) -> SceneParams:
    """The BASELINE config-#3 network (configs/scenes/toaster.ini shapes) with seeded
    weights on an analytic occupancy shape; defaults reproduce the survey's probe scene.
    arrays=False leaves out the two big arrays (table, bitfield: None) -- for ranks that receive them by broadcast."""
    rng = np.random.default_rng(seed)
    offsets, pls = hash_level_offsets()
    table = (np.random.default_rng(seed + 1).uniform(-table_scale, table_scale, size=(int(offsets[-1]), 2)).astype(np.float32)
             if arrays else None)
    ide_dim = (2 ** ide_deg - 1 + ide_deg) * 2
    mlps = {
        "sdf": make_mlp(rng, [32, 64, 64, 15]),
        "env": make_mlp(rng, [ide_dim, hidden_env, hidden_env, hidden_env, 12]),
        "diffuse": make_mlp(rng, [24, 32, 3]),
        "specular": make_mlp(rng, [28, 64, 64, 3]),
        "renv": make_mlp(rng, [4, 64, 64, 64, 12]),
    }
    mlps["sdf"][-1][1][0] = sdf_bias            # mean sdf slightly positive -> sigma ~ 1/(2 beta) scale
    mlps["specular"][-1][1][:] -= math.log(3)   # network.py:332: lower specular at init
    bitfield = occupancy_bitfield(shape or shell()) if arrays else None
    return SceneParams(bitfield=bitfield, offsets=offsets, per_level_scale=pls, table=table, mlps=mlps, beta=beta)


def lego_scene(table_scale: float = 0.1, shape=None, sdf_bias: float = 0.005, beta: float = 0.01, sh_degree: int = 4,
               seed: int = 0) -> SceneParams:
    """The BASELINE config-#2 network: hash grid + SDF MLP 32-64-64-15, diffuse MLP 12-32-3 on the
    geometry feature, specular MLP [SH(view dir), geo_feat, SH(normal), n.v] = 45-64-64-3; no environment network."""
    rng = np.random.default_rng(seed)
    offsets, pls = hash_level_offsets()
    table = np.random.default_rng(seed + 1).uniform(-table_scale, table_scale, size=(int(offsets[-1]), 2)).astype(np.float32)
    sh = sh_degree ** 2
    mlps = {
        "sdf": make_mlp(rng, [32, 64, 64, 15]),
        "diffuse": make_mlp(rng, [12, 32, 3]),
        "specular": make_mlp(rng, [sh + 12 + sh + 1, 64, 64, 3]),
    }
    mlps["sdf"][-1][1][0] = sdf_bias
    mlps["specular"][-1][1][:] -= math.log(3)
    return SceneParams(bitfield=occupancy_bitfield(shape or shell()), offsets=offsets, per_level_scale=pls, table=table,
                       mlps=mlps, beta=beta)



def plain_scene(table_scale: float = 0.02, shape=None, sdf_bias: float = 0.005, beta: float = 0.1, sh_degree: int = 4,
                seed: int = 4) -> SceneParams:
    """The plainest SDF configuration of the reference's network: hash grid + SDF MLP 32-64-64-15, diffuse MLP 12-32-3 on the geometry
    feature, specular MLP [SH(view dir), geo_feat] = 28-64-64-3 -- no environment network, no normal / n.v inputs.  What the torch-only
    render function (nerf/render_func/non_cuda_ray.py) can drive (tests/golden/plain_like.ini).  A smooth field on purpose (small table
    amplitudes, beta = init_beta = 0.1): the importance samples' positions depend on fp32 cumulative sums, and a density as steep as the other
    synthetic scenes' (beta = 0.01) would turn their last-bit differences between two torch backends into 1e-4 of the image."""
    rng = np.random.default_rng(seed)
    offsets, pls = hash_level_offsets()
    table = np.random.default_rng(seed + 1).uniform(-table_scale, table_scale, size=(int(offsets[-1]), 2)).astype(np.float32)
    mlps = {
        "sdf": make_mlp(rng, [32, 64, 64, 15]),
        "diffuse": make_mlp(rng, [12, 32, 3]),
        "specular": make_mlp(rng, [sh_degree ** 2 + 12, 64, 64, 3]),
    }
    mlps["sdf"][-1][1][0] = sdf_bias
    mlps["specular"][-1][1][:] -= math.log(3)
    return SceneParams(bitfield=occupancy_bitfield(shape or shell()), offsets=offsets, per_level_scale=pls, table=table,
                       mlps=mlps, beta=beta)


def sphere_table(xyz_encoding: np.ndarray, noise: float = 0.02, seed: int = 11) -> np.ndarray:
    """hash table of the env-sphere test scene: every level's rows hold that level's two features of the reference's `demo/
    xyz_encoding.txt` (the constant position feature its notebook distils the trained sphere's table into) plus seeded
    U(-noise, noise), level by level (numpy PCG64: platform independent) -- so that the SDF network shipped in demo/ sees the
    inputs it was trained on, perturbed enough for normals, densities and features to vary from sample to sample"""
    offsets, _ = hash_level_offsets()
    rng = np.random.default_rng(seed)
    table = np.empty((int(offsets[-1]), 2), np.float32)
    enc = np.asarray(xyz_encoding, np.float32).reshape(-1)
    for l in range(offsets.shape[0] - 1):
        n = int(offsets[l + 1] - offsets[l])
        table[offsets[l]:offsets[l + 1]] = enc[2 * l:2 * l + 2][None, :] + rng.uniform(-noise, noise, size=(n, 2)).astype(np.float32)
    return table
