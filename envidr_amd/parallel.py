"""Data-parallel rendering across the GPUs of one node: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for the tests).

The path shards naturally -- no sample of one ray depends on another ray -- so the
model state is replicated and only two things are distributed:

  * frames of a multi-view job (the env-rotation video of BASELINE config #5): view v goes to rank
    v % world (`views_for_rank`);
  * rays of a single frame: interleaved 8x8-pixel tiles (`tile_shard`), which balances hit / miss
    regions across ranks better than contiguous strips and keeps 64-ray wave locality.

The only collective is the gather of the finished image(s) to rank 0 (`gather_to_root`): 0.96 MB
of fp32 RGB per rank for an 800x800 frame split 8 ways, each sender using its own direct xGMI link.
The reference has no working multi-GPU path to mirror (its DDP leftovers are dead code,
nerf/utils.py:400-402,1353-1371).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None, force: bool = False, timeout_s: float | None = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; initialises the process group when
    WORLD_SIZE > 1 -- or, with `force`, also for a world of one (the collectives then run for real on a
    single rank: how the multi-GPU code path is exercised on a one-GPU box).  Rendezvous on 127.0.0.1
    unless MASTER_ADDR says otherwise.  `timeout_s` bounds the rendezvous (a missing rank then raises instead of hanging)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world == 1:
                import socket
                s = socket.socket()
                s.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s.getsockname()[1])
                s.close()
            else:
                os.environ["MASTER_PORT"] = "29500"
        # the host driver only supports dmabuf IPC; without this RCCL fails with hipIpcGetMemHandle: invalid argument
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        kw = {}
        if timeout_s is not None:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=timeout_s)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def preflight(rank: int, world: int, dev: torch.device, timeout_s: float = 30.0) -> dict:
    """The collectives a multi-GPU run depends on, once, small, under a time limit -- so that a node whose ranks cannot talk to each
    other yields a diagnosable record instead of a hang: one 1 MB `gather` to rank 0 (contents checked there) and the `ranks_seen`
    all-reduce.  Runs in a worker thread; the caller gets {"ok": bool, "stage": ..., "error": ..., "ms": {...}} and decides what to do
    (bench.py prints it as one JSON line and exits).  Needs an initialised process group."""
    import threading
    import time
    state = {"ok": False, "stage": "start", "error": None, "ms": {}, "rank": rank, "world": world}

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    def work():
        try:
            if dev.type == "cuda":
                torch.cuda.set_device(dev)
            state["stage"] = "gather_1MB"
            t0 = time.perf_counter()
            mine = torch.full((262144,), float(rank), dtype=torch.float32, device=dev)
            got = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
            dist.gather(mine, gather_list=got, dst=0)
            sync()
            if rank == 0:
                bad = [r for r, g in enumerate(got) if not bool((g == float(r)).all())]
                if bad:
                    raise RuntimeError(f"gather delivered wrong contents for rank(s) {bad}")
            state["ms"]["gather_1MB"] = (time.perf_counter() - t0) * 1e3
            state["stage"] = "all_reduce_ranks_seen"
            t0 = time.perf_counter()
            ones = torch.ones(1, dtype=torch.float32, device=dev)
            dist.all_reduce(ones, op=dist.ReduceOp.SUM)
            sync()
            seen = int(ones.item())
            state["ranks_seen"] = seen
            if seen != world:
                raise RuntimeError(f"all_reduce saw {seen} ranks of {world}")
            state["ms"]["all_reduce_ranks_seen"] = (time.perf_counter() - t0) * 1e3
            state["stage"] = "done"
            state["ok"] = True
        except Exception as e:      # noqa: BLE001
            state["error"] = f"{type(e).__name__}: {e}"[:500]

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        state["error"] = f"timed out after {timeout_s:.0f} s in {state['stage']} (a rank did not join the collective)"
    return dict(state)


def views_for_rank(num_views: int, rank: int, world: int) -> list[int]:
    """round-robin view assignment: rank r renders views r, r + world, ..."""
    return list(range(rank, num_views, world))


def tile_shard(H: int, W: int, rank: int, world: int, tile: int = 8) -> torch.Tensor:
    """flat pixel indices (row-major, int64) of the interleaved tile set owned by `rank`.
    Tiles are numbered row-major; tile t belongs to rank t % world.  Pixels inside a tile are listed
    row-major so each tile is one 64-ray wave's worth of neighbouring rays."""
    ty, tx = (H + tile - 1) // tile, (W + tile - 1) // tile
    tiles = torch.arange(ty * tx)
    mine = tiles[tiles % world == rank]
    y0 = (mine // tx) * tile
    x0 = (mine % tx) * tile
    dy, dx = torch.meshgrid(torch.arange(tile), torch.arange(tile), indexing="ij")
    ys = (y0[:, None] + dy.reshape(1, -1)).reshape(-1)
    xs = (x0[:, None] + dx.reshape(1, -1)).reshape(-1)
    keep = (ys < H) & (xs < W)
    return (ys[keep] * W + xs[keep]).to(torch.int64)


def shard_sizes(H: int, W: int, world: int, tile: int = 8) -> list[int]:
    return [int(tile_shard(H, W, r, world, tile).numel()) for r in range(world)]


def gather_to_root(local: torch.Tensor, sizes: list[int], root: int = 0) -> list[torch.Tensor] | None:
    """gather variable-length [n_r, C] tensors to `root`; returns the list on root, None elsewhere.
    Padded to the largest shard so that one fixed-size collective is issued (RCCL gather)."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    n_max = max(sizes)
    pad = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == root else None
    dist.gather(pad, gather_list=bufs, dst=root)
    if rank != root:
        return None
    return [b[:n] for b, n in zip(bufs, sizes)]


def assemble_frame(parts: list[torch.Tensor], H: int, W: int, tile: int = 8) -> torch.Tensor:
    """inverse of tile_shard on the root: scatter every rank's rays back to their pixels -> [H*W, C]."""
    world = len(parts)
    out = torch.empty((H * W,) + tuple(parts[0].shape[1:]), dtype=parts[0].dtype, device=parts[0].device)
    for r, p in enumerate(parts):
        out[tile_shard(H, W, r, world, tile).to(p.device)] = p
    return out


def render_frame_sharded(render_fn, rays_o: torch.Tensor, rays_d: torch.Tensor, H: int, W: int, key: str = "image",
                         tile: int = 8) -> torch.Tensor | None:
    """single-frame data parallelism: every rank renders its tile set with `render_fn(rays_o, rays_d)
    -> dict`, rank 0 returns the assembled [H*W, C] image, the others None."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return render_fn(rays_o, rays_d)[key]
    rank, world = dist.get_rank(), dist.get_world_size()
    idx = tile_shard(H, W, rank, world, tile).to(rays_o.device)
    local = render_fn(rays_o.view(-1, 3)[idx].contiguous(), rays_d.view(-1, 3)[idx].contiguous())[key]
    if local.dim() == 1:
        local = local[:, None]
    parts = gather_to_root(local, shard_sizes(H, W, world, tile))
    return assemble_frame(parts, H, W, tile) if parts is not None else None
