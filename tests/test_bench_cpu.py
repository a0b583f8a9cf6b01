"""bench.py's multi-rank plumbing on CPU: `python bench.py --gpus 2 --stub` must start its two ranks by itself (no torchrun),
partition the views, gather every frame to rank 0 and print ONE JSON line (gloo; a stand-in renderer)."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _run(args, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    # stdout carries the one JSON line and nothing else (library banners -- RCCL, Gloo -- are routed to stderr)
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    j = _run(["--gpus", "2", "--steps", "3", "--warmup", "2", "--stub"])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 2
    assert len(j["per_rank_ms_per_step"]) == 2 and j["scaling"] == "weak" and j["higher_is_better"] is True
    assert j["value"] > 0 and j["config"]["rays_per_step_per_gpu"] == 256


def test_single_rank_and_torchrun_style_environment():
    j = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--stub"])
    assert j["n_gpus"] == 1 and len(j["per_rank_ms_per_step"]) == 1
    # a rank launched by torch.distributed.run (environment already set) does not spawn again
    j = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--stub"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert j["n_gpus"] == 1


def test_strong_scaling_mode_shards_one_frame_by_tiles():
    """--scaling strong: every rank renders its interleaved 8x8-pixel tiles of ONE frame, rank 0 gathers (padded: the shards are
    ragged for 3 ranks on a 16x16 frame) and assembles; the stub's pixels carry their ray ids, checked inside bench.py"""
    for gpus in (2, 3):
        j = _run(["--gpus", str(gpus), "--steps", "3", "--warmup", "2", "--stub", "--scaling", "strong"])
        assert j["n_gpus"] == gpus and j["scaling"] == "strong" and j["dist"]["gathered_equals_rendered"] is True
        assert j["dist"]["ranks_seen"] == gpus                    # an all_reduce of ones over the benchmark's own process group
        assert j["config"]["rays_per_step_per_gpu"] in (128, 64) and j["value"] > 0


def test_force_dist_runs_one_rank_through_the_collective_path():
    j = _run(["--gpus", "1", "--steps", "3", "--warmup", "2", "--stub", "--force-dist"])
    assert j["n_gpus"] == 1 and j["dist"]["world"] == 1 and j["dist"]["force_dist"] is True and j["dist"]["gathered_equals_rendered"] is True
    assert j["dist"]["ranks_seen"] == 1
    j = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--stub", "--force-dist", "--scaling", "strong"])
    assert j["scaling"] == "strong" and j["dist"]["gathered_equals_rendered"] is True


def test_eight_ranks_the_drivers_largest_launch():
    """`python bench.py --gpus 8` as the driver's scaling run starts it (self-launched ranks), both modes, on gloo"""
    for mode in ("weak", "strong"):
        j = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--stub", "--scaling", mode])
        assert j["n_gpus"] == 8 and j["scaling"] == mode and len(j["per_rank_ms_per_step"]) == 8
        assert j["dist"]["gathered_equals_rendered"] is True and j["value"] > 0


def test_a_damaged_shard_of_any_rank_is_noticed():
    """dist.gathered_equals_rendered covers EVERY rank's payload: rank 0 renders the last step again and compares what the
    gather delivered for each sender.  --corrupt-rank damages one pixel of one rank's last image before it is sent."""
    for mode in ("weak", "strong"):
        j = _run(["--gpus", "3", "--steps", "3", "--warmup", "1", "--stub", "--scaling", mode])
        assert j["dist"]["gathered_equals_rendered"] is True and j["dist"]["gathered_equals_rendered_per_rank"] == [True, True, True]
        for bad in (1, 2, 0):
            j = _run(["--gpus", "3", "--steps", "3", "--warmup", "1", "--stub", "--scaling", mode, "--corrupt-rank", str(bad)])
            assert j["dist"]["gathered_equals_rendered"] is False
            assert j["dist"]["gathered_equals_rendered_per_rank"] == [r != bad for r in range(3)], (mode, bad, j["dist"])


def _run_failing(args, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    return json.loads(lines[0])


def test_preflight_reports_ok_and_rides_along_with_every_multi_rank_run():
    j = _run(["--gpus", "2", "--stub", "--preflight"])
    assert j["preflight"]["ok"] is True and j["preflight"]["ranks_seen"] == 2 and j["n_gpus"] == 2 and j["backend"] == "gloo"
    assert set(j["preflight"]["ms"]) == {"gather_1MB", "all_reduce_ranks_seen"}
    j = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--stub"])
    assert j["dist"]["preflight"]["ok"] is True and j["dist"]["preflight"]["stage"] == "done"


def test_preflight_failures_are_one_diagnosable_json_line_not_a_hang():
    # fewer devices than ranks (this container has no GPU at all): refused before any rank is started
    j = _run_failing(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert j["preflight"]["stage"] == "device_count" and j["preflight"]["ok"] is False and "device_count() = 0" in j["error"]
    # a rank that never reaches the rendezvous: rank 0 of a world of two, started alone
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    j = _run_failing(["--gpus", "2", "--stub", "--preflight"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                                              "MASTER_PORT": str(port), "ENVIDR_PREFLIGHT_TIMEOUT_S": "2"})
    assert j["preflight"]["stage"] == "init_process_group" and j["preflight"]["rank"] == 0
    # a rank that joins the process group but not the collectives: the others name the call they are stuck in
    j = _run_failing(["--gpus", "2", "--stub", "--preflight"], {"ENVIDR_PREFLIGHT_TIMEOUT_S": "3", "ENVIDR_PREFLIGHT_ABSENT_RANK": "1"})
    assert j["preflight"]["stage"] == "gather_1MB" and j["preflight"]["rank"] == 0 and "timed out" in j["preflight"]["error"]
