"""N > 1 path on CPU: world_size 2, gloo.  Exercises the ray / view partitioning and the image gather
(envidr_amd.parallel) with a stand-in render function (the HIP renderer needs a GPU; what is under
test here is the distribution logic, which is device independent)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from envidr_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_render(rays_o, rays_d):
    # deterministic per-ray "colour" so that assembly errors are visible
    return {"image": torch.stack([rays_o[:, 0] * 2 + rays_d[:, 1], rays_o[:, 1] - rays_d[:, 2], rays_o[:, 2] * rays_d[:, 0]], -1)}


def _worker(rank, world, port, H, W, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    rays_o = torch.rand(H * W, 3, generator=g)
    rays_d = torch.rand(H * W, 3, generator=g)
    img = parallel.render_frame_sharded(_fake_render, rays_o, rays_d, H, W)
    views = parallel.views_for_rank(7, rank, world)
    # multi-view job: each rank renders its views, root gathers them one by one
    frames = {}
    for v in range(0, 7, world):
        mine = v + rank
        local = _fake_render(rays_o + mine, rays_d)["image"] if mine < 7 else torch.zeros(H * W, 3)
        parts = parallel.gather_to_root(local, [H * W] * world)
        if rank == 0:
            for k, p in enumerate(parts):
                if v + k < 7:
                    frames[v + k] = p.clone()
    if rank == 0:
        q.put((img.numpy(), views, {k: f.numpy() for k, f in frames.items()}))
    else:
        assert img is None
        q.put((None, views, None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H,W", [(40, 56), (37, 29)])
def test_two_rank_sharded_frame_and_views(H, W):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, H, W, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    img = next(r[0] for r in results if r[0] is not None)
    frames = next(r[2] for r in results if r[2] is not None)
    g = torch.Generator().manual_seed(0)
    rays_o = torch.rand(H * W, 3, generator=g)
    rays_d = torch.rand(H * W, 3, generator=g)
    want = _fake_render(rays_o, rays_d)["image"].numpy()
    assert np.array_equal(img, want)                       # assembled frame == single-process render, bit for bit
    assert sorted(sum((r[1] for r in results), [])) == list(range(7))
    for v in range(7):
        assert np.array_equal(frames[v], _fake_render(rays_o + v, rays_d)["image"].numpy())


def test_tile_shard_is_a_partition():
    for H, W, world in [(800, 800, 8), (37, 29, 3), (8, 8, 2), (5, 5, 4)]:
        seen = torch.cat([parallel.tile_shard(H, W, r, world) for r in range(world)])
        assert seen.numel() == H * W and torch.equal(torch.sort(seen).values, torch.arange(H * W))
        sizes = parallel.shard_sizes(H, W, world)
        assert sum(sizes) == H * W
    # 800x800 over 8 ranks is perfectly balanced and every shard is whole 64-ray tiles
    assert set(parallel.shard_sizes(800, 800, 8)) == {80000}


def _scene_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.init_from_env(backend="gloo")
    import bench
    sc = bench.load_scene(rank, world, True, torch.device("cpu"))
    import hashlib
    q.put((rank, hashlib.sha256(sc.table.numpy().tobytes()).hexdigest(), hashlib.sha256(sc.bitfield.numpy().tobytes()).hexdigest(),
           float(sc.mlps["env"][1][0][3, 5]), tuple(sc.table.shape)))
    dist.barrier()
    dist.destroy_process_group()


def test_scene_is_generated_once_and_broadcast():
    """bench.load_scene: rank 0 generates the 48.8 MB table and the bitfield, the other ranks receive them by broadcast (and
    generate only the small MLPs from the seed): every rank ends up with the same scene as a locally generated one"""
    import hashlib
    import sys
    from pathlib import Path
    from envidr_amd import scenes
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scene_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = scenes.toaster_scene()
    th, bh = hashlib.sha256(want.table.tobytes()).hexdigest(), hashlib.sha256(want.bitfield.tobytes()).hexdigest()
    for rank, t, b, w, shape in results:
        assert (t, b) == (th, bh) and w == float(want.mlps["env"][1][0][3, 5]) and shape == want.table.shape, rank
    assert scenes.toaster_scene(arrays=False).table is None
