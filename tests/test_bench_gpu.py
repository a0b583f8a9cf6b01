"""The multi-GPU code path of bench.py on the ONE GPU a test box has: `--force-dist` puts a single rank through exactly what
N > 1 ranks run -- RCCL process group, the gather of every finished frame on a side stream, the slot events that keep an
output set from being overwritten before it has left, the closing barrier -- and bench.py itself compares what rank 0
received with what was rendered.  The plain single-rank run of the same workload must not be measurably faster."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _run(args):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    # stdout carries the one JSON line and nothing else (library banners -- RCCL, Gloo -- are routed to stderr)
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_force_dist_single_rank_rccl_path_matches_the_plain_run():
    plain = _run(["--gpus", "1", "--steps", "8", "--warmup", "2", "--headline-only"])
    forced = _run(["--gpus", "1", "--steps", "8", "--warmup", "2", "--headline-only", "--force-dist"])
    assert forced["dist"]["backend"] == "nccl" and forced["dist"]["world"] == 1 and forced["dist"]["ranks_seen"] == 1
    assert forced["dist"]["gathered_equals_rendered"] is True
    assert forced["config"]["samples_per_frame"] == plain["config"]["samples_per_frame"]
    assert forced["gather_ms"] > 0                                           # the gathers ran, on the side stream, and were timed
    # the gather overlaps the next frame: 0.1 - 1 % off the plain run when measured (profiles/r04*/bench_force_dist.json); the bound here only
    # catches a serialised gather (it would cost a frame's 7.7 MB copy + sync per step), not clock noise between two processes on a shared box
    assert forced["value"] >= 0.92 * plain["value"], (forced["value"], plain["value"])


def test_strong_scaling_mode_on_one_rank():
    """--scaling strong through the same collective path: tile shard of one rank = the whole frame in tile order, gathered and
    un-permuted on the side stream; the assembled frame equals the rendered one"""
    j = _run(["--gpus", "1", "--steps", "4", "--warmup", "2", "--headline-only", "--force-dist", "--scaling", "strong"])
    assert j["scaling"] == "strong" and j["dist"]["gathered_equals_rendered"] is True
    assert j["config"]["rays_per_step_per_gpu"] == 640000 and j["value"] > 5e6


def test_renderer_built_from_a_broadcast_scene_is_the_same_renderer():
    """ranks other than 0 receive the table and the bitfield as device tensors (bench.load_scene) and regenerate only the MLPs:
    the renderer built from that is bit-identical to one built from a locally generated scene"""
    import torch
    from envidr_amd import scenes
    from envidr_amd.fused import FusedRenderer
    full = scenes.toaster_scene()
    recv = scenes.toaster_scene(arrays=False)
    assert recv.table is None and recv.bitfield is None
    recv.table, recv.bitfield = torch.from_numpy(full.table).cuda(), torch.from_numpy(full.bitfield).cuda()
    a, b = FusedRenderer.from_scene(full), FusedRenderer.from_scene(recv)
    ro, rd = (torch.from_numpy(x).cuda() for x in scenes.camera_rays(64, 64))
    ra, rb = a.render_frame(ro, rd, 0.4, image_width=64), b.render_frame(ro, rd, 0.4, image_width=64)
    torch.cuda.synchronize()
    for k in ("image", "depth", "weights_sum", "normal_image", "diffuse_image", "specular_image"):
        assert torch.equal(ra[k], rb[k]), k
