#!/usr/bin/env python
"""Benchmark of the ENVIDR render hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: the script starts its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus 1 --force-dist                        (one rank through the N > 1 code path: RCCL world of 1)
    python bench.py --gpus N --scaling strong                    (ONE frame per step, its 8x8-pixel tiles interleaved over the ranks)

Metric (BASELINE.json): rendered rays/s at 800x800, max 1024 samples/ray, plus PSNR of the GPU image against the CPU
reference restatement of the same frame.

One "step" = every rank renders ONE full 800x800 view (640 000 primary rays) of the synthetic toaster scene -- march +
hash grid + SDF MLP + analytic normals (geometry pipeline), 2x IDE + 2x environment MLP + diffuse/specular heads (shading
pass), compositing, all four auxiliary images on (normal / diffuse / specular / roughness, as the reference's toaster.ini
renders them) -- and the finished RGB frame is gathered to rank 0 (RCCL gather over xGMI on a side stream, overlapped with
the next view; no-op at N = 1).  Views are the env-rotation video frames of BASELINE config #5: view v = step * N + rank
gets env rotation 2 pi v / 200, so per-GPU work is fixed as N grows (weak scaling).  Inputs (rays, table, weights) are
resident in HBM before the timed region.

The headline step renders the same camera every time, so the per-ray sample counts the pipeline keeps from the previous
frame are an EXACT hint (the frame is then one march round, nothing evaluated that is not composited).  Two more legs say
what that is worth (JSON keys `cold_frame`, `moving_camera`): every frame un-hinted, and a camera that moves 2 degrees per
step with `envidr_get_rays` inside the timed region and the previous pose's counts as the hint.

Printed by rank 0 as ONE JSON line; see DESIGN.md section "Measurement" for the roofline arithmetic.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 800
FLOP_PER_SAMPLE = 650_880          # dense-layer FLOPs per shaded sample, toaster network
FLOP_PER_SAMPLE_SHADING = 624_192  # of which in the shading pass: env MLP 305 152 x 2 + diffuse 1 728 + specular 12 160
FLOP_PER_SAMPLE_GEOMETRY = 26_688  # SDF network forward 14 208 + input gradient 12 480 (geometry pass)
FLOP_PER_SAMPLE_RELIGHT = 277_376  # neural_renderer.ini / shipped env nets: IDE 4, env hidden 160
FLOP_PER_SAMPLE_PLAIN = 41_984     # configs[1]: no env MLP
FLOP_PER_SAMPLE_RENV = 30_592      # third pass of indirect rendering: renv MLP 4-64-64-64-12 (18 432) + the specular head again (12 160)
HASH_BYTES_PER_SAMPLE = 1024       # 16 levels x 8 corners x 8 B gathered per sample
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32-input MFMA dense peak
PEAK_FP16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA peak
PEAK_HBM_GBPS = 8000.0
HEADLINE_SDF_BIAS = 0.065          # the scene density SURVEY.md 8(d) specifies: ~28 samples per ray at a 37 % hit rate (round 6: the headline)
HEADLINE_SIZING_HINT = 40.0        # samples per ray the headline's sample / record buffers are sized for before the first frame
PREFLIGHT_TIMEOUT_S = float(os.environ.get("ENVIDR_PREFLIGHT_TIMEOUT_S", "30"))      # (the tests shrink it)


def csrc_hash() -> str:
    """identifies the kernel sources a PMC summary was collected on (there is no .git on the GPU box)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "envidr_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".inc")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def host_cpu_info() -> dict:
    info = {"logical_cpus": os.cpu_count()}
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        for line in out.splitlines():
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k in ("Model name", "Socket(s)", "Core(s) per socket", "Thread(s) per core", "CPU max MHz", "CPU MHz", "Flags"):
                info[k] = v
    except Exception:
        pass
    return info


def host_peak_fp32_gflops(host: dict) -> dict | None:
    """what the host's cores could do in fp32 on paper: physical cores x 2 FMA pipes x vector lanes x 2 FLOP x clock (lscpu)"""
    try:
        cores = int(host["Socket(s)"]) * int(host["Core(s) per socket"])
        lanes = 16 if "avx512f" in host.get("Flags", "") else 8
        mhz = float(host.get("CPU max MHz") or host.get("CPU MHz") or 0.0)
        if mhz <= 0:
            return None
        return {"gflops": cores * 2 * lanes * 2 * mhz / 1e3, "formula": f"{cores} cores x 2 FMA pipes x {lanes} fp32 lanes x 2 FLOP x {mhz / 1e3:.2f} GHz"}
    except Exception:       # noqa: BLE001
        return None


def cpu_baseline(scene, env_rot: float, frames: int, res: int) -> dict:
    """the CPU restatement (oracle: C/OpenMP ops + torch CPU fp32 GEMMs, reference n_step schedule) timed on the host cores
    on a BOUNDED sample of the benchmark's workload: a `res` x `res` view (default 400: a quarter of the rays of the 800x800 frame, same scene,
    same camera, same samples per ray -- about 30 s of CPU work).  Thread count: every candidate in {16, 32, 64} (capped at the hardware
    threads) renders a 200x200 view of the SAME scene and camera once, after a small warm-up; the fastest is used."""
    from envidr_amd import scenes
    from oracle.py import render_oracle as ro
    opt = ro.RenderOptions(ide_mode="torch")
    probe_res = min(200, res)
    probe_o, probe_d = scenes.camera_rays(probe_res, probe_res)
    best, threads, tried = None, 1, {}
    ncpu = os.cpu_count() or 1
    # (all hardware threads is not a candidate: on the 2 x 64-core, 256-thread host of the GPU box that setting took 413 s for
    # a 400x400 frame the others render in 9-15 s -- profiles/r03g/bench.json)
    for th in sorted({min(c, ncpu) for c in (16, 32, 64)}):
        torch.set_num_threads(th)
        ro.render_rays(scene, probe_o[:256], probe_d[:256], opt, env_rot)    # warm-up (library loads, thread pools)
        t0 = time.perf_counter()
        ro.render_rays(scene, probe_o, probe_d, opt, env_rot)
        tried[th] = time.perf_counter() - t0
        if best is None or tried[th] < best:
            best, threads = tried[th], th
    torch.set_num_threads(threads)
    rays_o, rays_d = scenes.camera_rays(res, res)
    times, out = [], None
    for _ in range(max(frames, 1)):
        t0 = time.perf_counter()
        out = ro.render_rays(scene, rays_o, rays_d, opt, env_rot)
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    n = res * res
    host = host_cpu_info()
    try:
        host_cores = int(host["Socket(s)"]) * int(host["Core(s) per socket"])
    except Exception:       # noqa: BLE001
        host_cores = None
    peak = host_peak_fp32_gflops(host)
    host.pop("Flags", None)
    reached = out["n_samples"] * FLOP_PER_SAMPLE / dt / 1e9
    return {"value": n / dt, "unit": "rays/s", "cores": threads, "threads_used": threads, "host_cores": host_cores,
            "cores_note": "`cores` / `threads_used` = the torch + OpenMP thread count that rendered the probe view fastest, NOT the size of the "
                          "host (`host_cores` physical cores, `host.logical_cpus` hardware threads)",
            "kind": "port",
            "sample": f"{res}x{res} view of the same scene and camera ({n} rays, {out['n_samples']} samples, {out['n_samples'] / n:.1f} per ray as in the 800x800 "
                      f"frame): oracle/ C+OpenMP ops + torch CPU fp32 GEMMs, reference n_step schedule; median of {len(times)} view(s) ({dt:.1f} s each) at the "
                      f"fastest of the thread counts tried on a {probe_res}x{probe_res} view of the same scene",
            "frame_seconds": times, "threads_tried_seconds_on_probe_frame": tried, "probe_frame": f"{probe_res}x{probe_res}",
            "ide_mode": "torch (the reference's fp32 complex-power formulation, ide_encoder.py:98-130)", "host": host,
            "samples_per_s": out["n_samples"] / dt,
            "dense_layer_gflops_reached": reached, "host_fp32_peak_estimate": peak,
            "host_note": ("a reported baseline, not a target: the port keeps the reference's loop structure (per-round torch CPU GEMMs on the live samples, "
                          "complex-power IDE) and reaches the `dense_layer_gflops_reached` figure; `host_fp32_peak_estimate` is what the host's cores could do "
                          "on paper -- the ratio of the GPU figure to this baseline says nothing about kernel quality, `roofline.frac` does"),
            "image": out["image"]}


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawned(rank: int, world: int, port: int, argv: list[str]) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    run(argv)


def parse(argv: list[str]):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (profiling runs)")
    ap.add_argument("--cpu-frames", type=int, default=1, help="CPU views timed for cpu_baseline (median)")
    ap.add_argument("--cpu-res", type=int, default=400, help="resolution of the CPU view: a bounded sample of the 800x800 frame (same scene, camera and "
                    "samples per ray; 400 = a quarter of its rays, about 30 s of CPU work)")
    ap.add_argument("--preflight", action="store_true", help="multi-GPU pre-flight only: device count, process group, one 1 MB gather and the ranks_seen "
                    "all-reduce under a 30 s limit; prints one JSON line (ok or which rank / which call failed) and exits.  Runs automatically before the "
                    "timed loop of every N > 1 run")
    ap.add_argument("--path", choices=["pipeline", "fused"], default="pipeline",
                    help="pipeline: geometry pipeline + shading pass per frame (default); fused: one persistent kernel per frame")
    ap.add_argument("--headline-only", action="store_true", help="skip the CPU leg and the other_configs renders, so that a "
                    "profiler sees only the headline kernels' launches")
    ap.add_argument("--force-dist", action="store_true", help="run a single rank through the N > 1 code path: process group of one "
                    "(RCCL on the GPU), gather on the side stream, slot events, barrier -- the branch the multi-GPU run takes")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="weak: one full view per rank per step (default); "
                    "strong: ONE view per step, its 8x8-pixel tiles interleaved over the ranks, gathered and assembled on rank 0")
    ap.add_argument("--hinted", action="store_true", help="timed frames WITH the previous frame's per-ray sample counts as a hint (the fixed-camera "
                    "video of BASELINE configs[4]); the default is the cold frame: configs[2] is ONE view, nothing is known about it beforehand")
    ap.add_argument("--cold", action="store_true", help="(the default since round 5; kept so that older command lines still parse)")
    ap.add_argument("--corrupt-rank", type=int, default=-1, help="test hook: this rank damages one pixel of the LAST step's image before "
                    "sending it; dist.gathered_equals_rendered must then come out false for that rank (tests/test_bench_cpu.py)")
    ap.add_argument("--stub", action="store_true", help="CPU plumbing test: gloo, a stand-in renderer, tiny frames "
                    "(tests/test_bench_cpu.py); exercises launch, view partition, overlapped gather and the JSON line")
    return ap.parse_args(argv)


class StubRenderer:
    """stand-in for FusedRenderer in --stub mode: the frame of view v is the constant v (so the root can check what it gathered)"""

    def render_frame(self, rays_o, rays_d, env_rot, out=None, events=None, wait=True, **kw):
        n = rays_o.shape[0]
        res = out if out is not None else {}
        # strong scaling: every pixel carries its own ray id as well, so the assembled frame can be checked pixel by pixel
        res["image"] = torch.stack([torch.full((n,), float(env_rot)), rays_o[:, 0], rays_d[:, 0]], -1)
        res["n_records"] = res["n_samples"] = 12 * n
        return res

    def check_frames(self):
        pass


def load_scene(rank: int, world: int, dist_on: bool, dev):
    """The synthetic scene on every rank.  Rank 0 generates it (table 48.8 MB + bitfield: seconds of host time); with more than
    one rank the two big arrays are broadcast (RCCL over xGMI: "model state replicated, broadcast once at load")
    instead of every rank generating its own copy; the MLP weights (< 1 MB, milliseconds) are generated from the seed everywhere."""
    import torch.distributed as dist
    from envidr_amd import scenes
    if not dist_on:
        return scenes.toaster_scene(sdf_bias=HEADLINE_SDF_BIAS)
    # (a forced world of one goes through the same calls: the broadcasts are then trivial, but the code that runs on N ranks has run)
    sc = scenes.toaster_scene(sdf_bias=HEADLINE_SDF_BIAS, arrays=(rank == 0))
    rows = int(sc.offsets[-1])
    table = torch.from_numpy(sc.table).to(dev) if rank == 0 else torch.empty(rows, 2, dtype=torch.float32, device=dev)
    bitfield = torch.from_numpy(sc.bitfield).to(dev) if rank == 0 else torch.empty(sc.cascades * sc.grid_size ** 3 // 8, dtype=torch.uint8, device=dev)
    dist.broadcast(table, src=0)
    dist.broadcast(bitfield, src=0)
    sc.table, sc.bitfield = table, bitfield
    return sc


def run(argv: list[str]) -> None:
    args = parse(argv)
    args.cold = not args.hinted
    from envidr_amd import parallel
    import torch.distributed as dist

    stub = args.stub
    # stdout of a distributed run carries the ONE JSON line of rank 0 and nothing else: the communication libraries print
    # banners there (RCCL when its first communicator comes up, Gloo at rendezvous), so file descriptor 1 is pointed away before
    # the process group exists -- at stderr on rank 0 (until _finish() restores it for the JSON line), at /dev/null elsewhere
    global _REAL_STDOUT
    sys.stdout.flush()
    if int(os.environ.get("RANK", "0")) != 0:
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    elif int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.force_dist:
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)
    env_rank, env_world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    will_dist = env_world > 1 or args.force_dist
    if not stub and will_dist and torch.cuda.device_count() < env_world:
        _preflight_failed({"ok": False, "stage": "device_count", "rank": env_rank, "world": env_world,
                           "error": f"torch.cuda.device_count() = {torch.cuda.device_count()} < {env_world} ranks"})
    try:
        rank, world, local = parallel.init_from_env(backend="gloo" if stub else None, force=args.force_dist,
                                                    timeout_s=4 * PREFLIGHT_TIMEOUT_S if will_dist else None)
    except Exception as e:      # noqa: BLE001
        _preflight_failed({"ok": False, "stage": "init_process_group", "rank": env_rank, "world": env_world, "error": f"{type(e).__name__}: {e}"[:500],
                           "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}"})
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist_on = world > 1 or args.force_dist         # the collective code path (a world of one with --force-dist)
    strong = args.scaling == "strong"
    if stub:
        dev = torch.device("cpu")
        n_side = 16 if world <= 4 else 8 * int(np.ceil(np.sqrt(world)))           # at least one 8x8 tile per rank
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        n_side = H
    # multi-GPU pre-flight: the collectives the run depends on, small and under a time limit, before anything expensive
    pre = None
    if dist_on:
        if os.environ.get("ENVIDR_PREFLIGHT_ABSENT_RANK") == str(rank):      # test hook: this rank never joins the pre-flight collectives
            time.sleep(3 * PREFLIGHT_TIMEOUT_S)
            os._exit(4)
        pre = parallel.preflight(rank, world, dev, PREFLIGHT_TIMEOUT_S)
        if not pre["ok"]:
            _preflight_failed(pre)
    if args.preflight:
        _finish({"preflight": pre if pre is not None else {"ok": True, "stage": "done", "note": "one rank, no process group: nothing to check"},
                 "n_gpus": world, "backend": dist.get_backend() if dist_on else None} if rank == 0 else None, dist_on)
        return
    N_frame = n_side * n_side

    def env_rot(view: int) -> float:
        return 2 * math.pi * (view % 200) / 200

    if stub:
        scene, renderer = None, StubRenderer()
        rays_o = torch.arange(N_frame, dtype=torch.float32)[:, None].expand(N_frame, 3).contiguous()
        rays_d = rays_o * 2
    else:
        from envidr_amd import scenes
        from envidr_amd.fused import FusedRenderer
        scene = load_scene(rank, world, dist_on, dev)
        renderer = FusedRenderer.from_scene(scene, device=dev)
        rays_o, rays_d = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(H, W))
    # strong scaling: this rank's interleaved 8x8-pixel tiles of THE frame; weak: the whole frame (its own view)
    sizes = parallel.shard_sizes(n_side, n_side, world) if strong else [N_frame] * world
    full_rays = (rays_o, rays_d)                 # rank 0 re-renders what the others sent after the timed region (all N_frame rays)
    if strong:
        mine = parallel.tile_shard(n_side, n_side, rank, world).to(dev)
        rays_o, rays_d = rays_o[mine].contiguous(), rays_d[mine].contiguous()
        place = torch.cat([parallel.tile_shard(n_side, n_side, r, world) for r in range(world)]).to(dev) if rank == 0 else None
    N = rays_o.shape[0]                          # rays this rank renders per step
    n_max = max(sizes)
    pipeline = args.path == "pipeline"
    # two sets of output images: the gather of view i runs on a side stream while view i + 1 is rendered into the other set
    outs = [{}, {}]
    ray_cost = None if (stub or pipeline) else torch.zeros(N, dtype=torch.int16, device=dev)
    gather_lists = [[torch.empty(n_max, 3, device=dev) for _ in range(world)] if (dist_on and rank == 0) else None for _ in range(2)]
    send_pad = [torch.zeros(n_max, 3, device=dev) for _ in range(2)] if (dist_on and N != n_max) else None    # ragged shards: padded sends
    frames = [torch.empty(N_frame, 3, device=dev) for _ in range(2)] if (strong and rank == 0) else None
    comm = torch.cuda.Stream(dev) if (dist_on and not stub) else None
    gather_ev = []          # (start, end) events of the gathers on the side stream

    def frame(view: int, slot: int, events=None):
        if pipeline:
            # (weak mode: the rays ARE a row-major 800-wide image -- the layout hint lets the pipeline form its 64-ray blocks from
            #  8x8-pixel tiles; strong mode already lists a rank's rays tile by tile)
            return renderer.render_frame(rays_o, rays_d, env_rot(view) if not stub else float(view), out=outs[slot], events=events, wait=False,
                                         use_cost_hint=not args.cold, image_width=0 if strong else n_side, samples_per_ray_hint=HEADLINE_SIZING_HINT)
        if events:
            events[0].record()
        res = renderer.render(rays_o, rays_d, env_rot(view), extras=True, stats=True, out=outs[slot], ray_cost=ray_cost)
        if events:
            events[1].record()
        return res

    slot_sent = [None, None]     # event on the side stream: the gather that reads this output set has finished

    last_step = args.warmup + args.steps - 1

    def deliver(res, slot, timed, i):
        """the finished image -> rank 0 (+ assembly of the tile shards there); runs on the side stream when there is one"""
        img = res["image"]
        if i == last_step and rank == args.corrupt_rank:
            img = img.clone()
            img[0, 0] += 1.0
        if send_pad is not None:
            send_pad[slot][:N].copy_(img)
            img = send_pad[slot]
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if (timed and comm is not None) else None
        if ev: ev[0].record()
        dist.gather(img, gather_list=gather_lists[slot], dst=0)
        if strong and rank == 0:
            # un-permutation of the tiles: one indexed copy over the concatenated shards
            frames[slot][place] = torch.cat([g[:n] for g, n in zip(gather_lists[slot], sizes)])
        if ev:
            ev[1].record()
            gather_ev.append(ev)

    def step(i: int, events=None, timed=False):
        slot = i & 1
        if slot_sent[slot] is not None:
            torch.cuda.current_stream(dev).wait_event(slot_sent[slot])     # (two steps ago; the gather of the last step keeps running)
        res = frame(i if strong else i * world + rank, slot, events)
        if dist_on:
            if comm is None:
                deliver(res, slot, timed, i)
            else:
                done = torch.cuda.Event()
                done.record()
                with torch.cuda.stream(comm):
                    comm.wait_event(done)
                    deliver(res, slot, timed, i)
                    slot_sent[slot] = torch.cuda.Event()
                    slot_sent[slot].record()
        return res

    def fence():
        if comm is not None:
            torch.cuda.current_stream(dev).wait_stream(comm)
        if dist_on:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize(dev)

    res = None
    for i in range(args.warmup):
        res = step(i)
    if dist_on:
        fence()
        _flush_c_stdio()          # the communicator exists now: its start-up banner goes out here, not at exit

    # HIP events on the stream the kernels are launched on (torch's current stream), recorded at the pass boundaries
    ev = None if stub else [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        res = step(args.warmup + i, ev[i] if ev else None, timed=True)
    fence()
    dt_local = time.perf_counter() - t0
    renderer.check_frames()
    last = last_step
    delivered_ok = delivered_per_rank = None
    if dist_on and rank == 0 and args.steps > 0:
        # What arrived is what every sender should have rendered -- for EVERY rank, not only the root's own shard: rank 0 renders
        # the last step again itself (weak: rank r's view, one frame each; strong: the whole frame, all ranks' tiles) into fresh
        # buffers and requires the gathered pixels to be bit-identical (rays are independent and the kernels deterministic: a
        # rank's shard of a frame equals those rays of the whole frame, tests/test_fullsize_gpu.py).
        def render_again(view):
            ro, rd = full_rays
            if pipeline:
                r2 = renderer.render_frame(ro, rd, env_rot(view) if not stub else float(view), out={}, wait=False,
                                           use_cost_hint=False, image_width=n_side, samples_per_ray_hint=HEADLINE_SIZING_HINT)
            else:
                r2 = renderer.render(ro, rd, env_rot(view), extras=True, stats=True, out={})
            return r2["image"]
        delivered_per_rank = []
        if strong:
            want = render_again(last)
            first = 0
            for r in range(world):
                shard = place[first:first + sizes[r]]
                first += sizes[r]
                delivered_per_rank.append(bool(torch.equal(gather_lists[last & 1][r][:sizes[r]], want[shard])))
            assembled = bool(torch.equal(frames[last & 1], want))                    # ... and the tiles went to the right pixels
            delivered_ok = assembled and all(delivered_per_rank)
        else:
            for r in range(world):
                delivered_per_rank.append(bool(torch.equal(gather_lists[last & 1][r][:N], render_again(last * world + r))))
            delivered_ok = all(delivered_per_rank)
        if not stub:
            renderer.check_frames()
    samples = 0 if res is None else int(res.get("n_records", 0))
    geometry_ms = kernel_ms = composite_ms = 0.0
    if not stub:
        if pipeline:
            geometry_ms, kernel_ms, composite_ms = (float(np.mean([e[j].elapsed_time(e[j + 1]) for e in ev])) for j in range(3))
            samples = int(renderer._frame["last"][1])
            evaluated = int(renderer._frame["last"][0])
        else:
            kernel_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
            samples = evaluated = int(outs[last & 1]["stats"][0].item())
    else:
        evaluated = samples
    gather_ms = float(np.mean([a.elapsed_time(b) for a, b in gather_ev])) if gather_ev else 0.0
    t = torch.tensor([dt_local], device=dev, dtype=torch.float64)
    per_rank = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)] if dist_on else [t]
    if dist_on:
        dist.all_gather(per_rank, t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    ranks_seen = None
    if dist_on:
        ones = torch.ones(1, device=dev, dtype=torch.float32)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)          # how many ranks the collective library really joined
        ranks_seen = int(ones.item())

    if rank == 0:
        rays_per_step = N_frame if strong else world * N_frame
        rays_per_s = rays_per_step * args.steps / dt
        flops = samples * (FLOP_PER_SAMPLE_SHADING if pipeline else FLOP_PER_SAMPLE) / max(kernel_ms * 1e-3, 1e-12) / 1e12
        if strong:
            par = (f"ONE 800x800 view per step, its 8x8-pixel tiles interleaved over {world} rank(s) (parallel.tile_shard), RCCL gather of the "
                   "shards + one indexed copy into the frame on rank 0, on a side stream (overlapped with the next view)")
        else:
            par = f"views x{world} + RCCL image gather on a side stream (overlapped with the next view)"
        if args.force_dist and world == 1:
            par += "; --force-dist: a process group of ONE rank through the same gather / event / barrier code"
        result = {
            "metric": "rendered rays/s at 800x800, 1024 max samples/ray", "value": rays_per_s, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # (the first ~120 characters carry what a truncated record must still show: the scene's samples per ray -- rays/s scales
            #  inversely with it -- and whether the frames had the previous frame's per-ray counts as a hint)
            # (the first ~120 characters carry what a truncated record must still show: COLD or HINTED, the scene's samples per ray --
            #  rays/s scales inversely with it -- and samples/s; at N = 1 the SURVEY 8(d)-density leg is spliced in behind them below)
            "config": {"workload": ("COLD" if args.cold else "HINTED (fixed-camera video)") + f" 800x800 toaster.ini frames at SURVEY 8d density, {samples / max(N, 1):.2f} samples/ray, "
                                   f"{(1 if strong else world) * samples * args.steps / dt / 1e6:.0f} M samples/s; "
                                   + ("no per-ray hint; " if args.cold else "per-ray counts of the previous frame as hint; ")
                                   + f"synthetic shell, toaster_scene(sdf_bias={HEADLINE_SDF_BIAS}); BASELINE configs[2]/[4]: hash L16xC2 + SDF 32-64-64-15 + IDE deg5 + env MLP "
                                   "72-256-256-256-12 x2 + diffuse/specular heads, one view per "
                                   + ("step sharded by 8x8-pixel tiles" if strong else "GPU per step, env-rotation video frames sharded by view")
                                   + ", normal/diffuse/specular/roughness images on",
                       "rays_per_step_per_gpu": N, "samples_per_frame": samples, "samples_evaluated_per_frame": evaluated,
                       "samples_per_ray": samples / max(N, 1), "max_steps": 1024, "T_thresh": 1e-4, "parallelism": par,
                       "schedule": ("geometry pipeline (device-driven march rounds + sample-parallel hash grid / SDF network, one record "
                                    "per composited sample) -> shading pass (k_shade_samples) -> per-ray composite; every frame marches, "
                                    "evaluates and shades from scratch.  "
                                    + ("NO per-ray hint (the default): 16 samples first, then chunks predicted per ray from its transmittance; "
                                       "video_fixed_camera is the same loop with the previous frame's counts as hint."
                                       if args.cold else
                                       "The camera is fixed (env-rotation video), so the per-ray sample counts kept from the previous frame "
                                       "are an EXACT hint: one march round, samples_evaluated == samples_composited (--hinted).")
                                    if pipeline else "one persistent kernel per frame")},
            "samples_per_s": (1 if strong else world) * samples * args.steps / dt,
            "per_rank_ms_per_step": [float(x.item()) / args.steps * 1e3 for x in per_rank],
            "gather_ms": gather_ms,
        }
        if dist_on:
            result["dist"] = {"backend": dist.get_backend(), "world": world, "ranks_seen": ranks_seen, "preflight": pre,
                              "ranks_seen_note": "all_reduce(SUM) of a one per rank over the benchmark's own process group",
                              "force_dist": bool(args.force_dist), "gathered_equals_rendered": delivered_ok,
                              "gathered_equals_rendered_per_rank": delivered_per_rank,
                              "gathered_check": "rank 0 re-rendered every rank's last view / the whole last frame and compared the gathered pixels bit for bit",
                              "scene": "none (stub)" if stub else ("generated on rank 0, table + bitfield broadcast" if dist_on else "generated locally")}
        if stub:
            result["config"]["workload"] = "STUB (CPU plumbing test)"
            _finish(result, dist_on)
            return
        result["roofline"] = {
            "bound": "mfma", "achieved": flops, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": flops / PEAK_FP32_MFMA_TFLOPS,
            "traffic": None, "kernel": "k_shade_samples<5,8>" if pipeline else "k_render_persistent<5,8,0>", "kernel_ms": kernel_ms,
            "kernel_ms_note": ("HIP events around the shading launch (envidr_shade_records) on its stream" if pipeline else
                               "HIP events around one envidr_render_rays call on its stream: the persistent kernel plus its two "
                               "pre-pass kernels"),
            "algorithmic_flop_per_sample": FLOP_PER_SAMPLE_SHADING if pipeline else FLOP_PER_SAMPLE, "samples_per_launch": samples,
            "hbm_view": {"bound": "hbm", "achieved": samples * HASH_BYTES_PER_SAMPLE / (kernel_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS,
                         "unit": "GB/s"}}
        if pipeline:
            frame_s = float(per_rank[0].item()) / args.steps         # rank 0's own frame time (its shard in strong mode)
            result["frame"] = {
                "geometry_ms": geometry_ms, "shading_ms": kernel_ms, "composite_ms": composite_ms,
                "whole_frame_mfma_TFLOPs": samples * FLOP_PER_SAMPLE / frame_s / 1e12,
                "whole_frame_mfma_frac": samples * FLOP_PER_SAMPLE / frame_s / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                # the geometry pass is the HBM-side regime of the path: gathers + SDF network
                "geometry_pass_roofline": {"hbm": {"achieved": evaluated * HASH_BYTES_PER_SAMPLE / (geometry_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS,
                                                   "unit": "GB/s", "frac": evaluated * HASH_BYTES_PER_SAMPLE / (geometry_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS},
                                           "mfma": {"achieved": evaluated * FLOP_PER_SAMPLE_GEOMETRY / (geometry_ms * 1e-3) / 1e12,
                                                    "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                                    "frac": evaluated * FLOP_PER_SAMPLE_GEOMETRY / (geometry_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS}},
                "geometry_pass_bound_note": ("neither roofline binds this stage: k_geo_eval32 runs at the SUM of its matrix-pipe time (208 fp32 MFMA steps' worth "
                                             "per 32 samples, about two thirds) and its vector issue time (1 718 instructions per batch, about one third) -- on this "
                                             "chip a wave issuing MFMAs back to back starves its SIMD partner whatever their priorities (tools/probe/"
                                             "cross_wave_probe.hip), so two waves per SIMD do not overlap the two; without any table load the kernel was 8 % faster, "
                                             "halving its L2 misses (tile order) bought 2 %, dropping the selects of the paired gathers 8 % (DESIGN.md 3.1)"),
                "record_bytes_per_sample": 92}
        if world == 1 and not strong and not args.headline_only and pipeline:
            context_legs(result, renderer, dev, args.steps, N_frame, headline_cold=args.cold)
        if world == 1 and not strong and not args.headline_only:
            other_configs(result, dev, rays_o, rays_d, N)
            training_and_loop_legs(result)
        # HBM traffic of the dominant kernel from the PMC summary -- only if it was collected on THESE kernel sources
        prof = os.path.join(ROOT, "profiles", "pmc_latest.json")
        note = "no profiles/pmc_latest.json"
        if os.path.exists(prof):
            try:
                p = json.load(open(prof))
                if p.get("csrc_sha") == csrc_hash():
                    result["roofline"]["traffic"] = p.get("hbm_bytes_per_launch")
                    note = f"profiles/pmc_latest.json ({p.get('tag', '?')}), FETCH_SIZE + WRITE_SIZE of {p.get('dominant_kernel', '?')} per launch"
                else:
                    note = (f"STALE: profiles/pmc_latest.json was collected on kernel sources {p.get('csrc_sha')}, these are {csrc_hash()} "
                            "-- rerun tools/gpu_round.sh")
                    print("bench.py: " + note, file=sys.stderr)
            except Exception as e:      # noqa: BLE001
                note = f"unreadable pmc_latest.json: {e}"
        result["roofline"]["traffic_note"] = note
        if not args.no_cpu_baseline and not args.headline_only and world == 1 and not strong:
            from envidr_amd import scenes
            cpu = cpu_baseline(scene, env_rot(0), args.cpu_frames, args.cpu_res)
            so, sd = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(args.cpu_res, args.cpu_res))
            g = renderer.render_frame(so, sd, env_rot(0), samples_per_ray_hint=HEADLINE_SIZING_HINT)["image"].cpu().numpy()
            ref = cpu.pop("image")
            mse = float(np.mean((g.astype(np.float64) - ref) ** 2))
            result["psnr_vs_cpu_reference_db"] = -10 * math.log10(max(mse, 1e-20))
            result["rel_l2_vs_cpu_reference"] = float(np.linalg.norm(g.astype(np.float64) - ref) / np.linalg.norm(ref))
            result["rel_l2_note"] = ("GPU side: IDE as fp64 Horner on the reference's fp32 coefficient table (csrc/ide_encoder.hip); CPU side: "
                                     "ide_mode='torch', the reference's fp32 complex-power formulation, whose own l = 16 terms carry ~1e-2 of "
                                     "rounding noise (DESIGN.md 4.4) -- most of this figure; with ide_mode='exact' on the CPU the two sides "
                                     "agree to ~1e-7 (tests/test_geometry_gpu.py::test_integer_trace_is_the_oracles)")
            result["cpu_baseline"] = cpu
        _finish(result, dist_on)
        return
    _finish(None, dist_on)


_REAL_STDOUT = None          # rank 0 of a distributed run: the saved stdout (see run())


def _preflight_failed(info: dict) -> None:
    """one diagnosable JSON line -- which rank, which call -- on stderr (every rank) and on the real stdout (rank 0), then leave
    without waiting for a communicator that may never come back"""
    line = json.dumps({"preflight": info, "error": f"multi-GPU pre-flight failed on rank {info.get('rank')} in {info.get('stage')}: {info.get('error')}"})
    print(line, file=sys.stderr, flush=True)
    if int(info.get("rank", 0)) == 0:
        try:
            os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (line + "\n").encode())
        except OSError:
            pass
    os._exit(3)


def _flush_c_stdio() -> None:
    """RCCL writes a version banner with C stdio (buffered until the process exits, i.e. AFTER anything Python has printed):
    push it out now so that the JSON line is the last thing on stdout"""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:       # noqa: BLE001
        pass


def _finish(result, dist_on: bool) -> None:
    """tear the process group down first (whatever the communication library still has to say comes out here), then rank 0
    prints the ONE JSON line, last"""
    import torch.distributed as dist
    if dist_on and dist.is_initialized():
        dist.destroy_process_group()
    _flush_c_stdio()
    sys.stdout.flush()
    global _REAL_STDOUT
    if _REAL_STDOUT is not None:
        os.dup2(_REAL_STDOUT, 1)
        os.close(_REAL_STDOUT)
        _REAL_STDOUT = None
    if result is not None:
        print(json.dumps(result), flush=True)


def context_legs(result, renderer, dev, steps: int, N: int, headline_cold: bool = True) -> None:
    """the headline workload in its other regimes: (a) the loop the headline did NOT run -- `video_fixed_camera` (BASELINE configs[4]: the
    previous frame's per-ray counts are an exact hint) when the headline is cold, `cold_frame` when it was run --hinted; (b) a camera
    that moves 2 degrees per step, rays generated on the device inside the timed region, hint = the previous pose's counts; (c) the same
    network on the density SURVEY.md 8(d) specifies (~28 samples per ray), cold like the headline"""
    from envidr_amd import scenes
    from envidr_amd.nerf.utils import get_rays
    steps = max(steps, 4)

    def leg(make_rays, use_hint: bool, warm: int, sizing_hint: float = HEADLINE_SIZING_HINT) -> dict:
        out: dict = {}
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
        log = []
        # one blocking frame first: it sizes the renderer's sample / record buffers (a frame that does not fit is rendered again
        # with larger ones), so that none of the asynchronous frames below can overflow
        renderer.render_frame(*make_rays(0), 0.1, out=out, wait=True, use_cost_hint=False, image_width=W, samples_per_ray_hint=sizing_hint)
        renderer.frame_log = {}
        for i in range(warm):
            o, d = make_rays(i)
            renderer.render_frame(o, d, 0.1, out=out, wait=False, use_cost_hint=use_hint, image_width=W)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            o, d = make_rays(warm + i)
            renderer.render_frame(o, d, 0.1, out=out, events=ev[i], wait=False, use_cost_hint=use_hint, image_width=W)
            log.append(renderer.frame_log[""])
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / steps
        renderer.check_frames()
        renderer.frame_log = None
        stats = torch.stack(log).cpu().numpy()
        geo, shade, comp = (float(np.mean([e[j].elapsed_time(e[j + 1]) for e in ev])) for j in range(3))
        return {"ms_per_frame": dt * 1e3, "rays_per_s": N / dt, "geometry_ms": geo, "shading_ms": shade, "composite_ms": comp,
                "samples_evaluated_per_frame": float(stats[:, 0].mean()), "samples_composited_per_frame": float(stats[:, 1].mean()),
                "evaluated_over_composited": float(stats[:, 0].sum() / max(stats[:, 1].sum(), 1)), "frames": steps}

    fixed = tuple(torch.from_numpy(a).to(dev) for a in scenes.camera_rays(H, W))
    if headline_cold:
        result["video_fixed_camera"] = dict(leg(lambda i: fixed, True, 2),
                                            note="BASELINE configs[4], one GPU's share: env-rotation video frames of a FIXED camera -- the per-ray sample counts "
                                                 "kept from the previous frame are an exact hint (one march round, evaluated == composited); every frame "
                                                 "still marches, evaluates and shades from scratch")
    else:
        result["cold_frame"] = dict(leg(lambda i: fixed, False, 1),
                                    note="use_cost_hint=False every frame: 16 samples per ray first, then per-ray predicted chunks")
    # moving camera: theta advances 2 degrees per step; poses uploaded once, rays by envidr_get_rays INSIDE the timed region
    n_pose = steps + 2
    poses = torch.from_numpy(np.stack([scenes.nerf_matrix_to_ngp(scenes.pose_spherical(30.0 + 2.0 * i, -20.0, 4.0), scale=0.65)
                                       for i in range(n_pose)]).astype(np.float32)).to(dev)
    intr = scenes.intrinsics_for(H, W)

    def moving(i):
        r = get_rays(poses[i % n_pose:i % n_pose + 1], intr, H, W)
        return r["rays_o"][0], r["rays_d"][0]
    result["moving_camera"] = dict(leg(moving, True, 2),
                                   note="camera orbit advanced 2 degrees per frame; envidr_get_rays inside the timed region; the per-ray hint is "
                                        "the previous pose's sample counts (wrong for rays near silhouettes: extra rounds / zero-filled slots)")
    # rays/s scales inversely with the scene's samples per ray.  Rounds 1-5 quoted `value` on a thinner shell (sdf_bias = 0.005: ~12 samples
    # per ray at the same 37 % hit rate); since round 6 the headline is the density SURVEY.md 8(d) specifies and the thin shell is this
    # context leg: the portable figure is samples/s, which should not move between the two.
    from envidr_amd.fused import FusedRenderer
    headline_renderer = renderer
    renderer = FusedRenderer.from_scene(scenes.toaster_scene(), device=dev)
    thin = leg(lambda i: fixed, not headline_cold, 2, sizing_hint=20.0)
    renderer = headline_renderer
    result["thin_shell_scene"] = dict(thin, samples_per_ray=thin["samples_composited_per_frame"] / N,
                                      samples_per_s=thin["samples_composited_per_frame"] / (thin["ms_per_frame"] * 1e-3),
                                      note="toaster_scene(sdf_bias=0.005): the thinner shell rounds 1-5 quoted `value` on (~12 samples per ray), "
                                           + ("cold" if headline_cold else "hinted") + " frames like the headline")
    d = result["thin_shell_scene"]
    head, sep, tail = result["config"]["workload"].partition(" M samples/s; ")
    result["config"]["workload"] = (head + sep + f"on the thinner shell of rounds 1-5 ({d['samples_per_ray']:.1f} samples/ray): {d['rays_per_s'] / 1e6:.2f} M rays/s, "
                                    f"{d['samples_per_s'] / 1e6:.0f} M samples/s; " + tail)


def _time(fn, reps: int, dev) -> float:
    fn()
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t1) / reps


def _both_rooflines(samples: int, seconds: float, flop_per_sample: int) -> dict:
    hbm = samples * HASH_BYTES_PER_SAMPLE / seconds / 1e9
    mfma = samples * flop_per_sample / seconds / 1e12
    return {"hbm": {"achieved": hbm, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": hbm / PEAK_HBM_GBPS},
            "mfma": {"achieved": mfma, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": mfma / PEAK_FP32_MFMA_TFLOPS}}


def other_configs(result, dev, rays_o, rays_d, N) -> None:
    """the other single-GPU BASELINE configurations on the same camera (parity-test configurations, not the headline)"""
    from envidr_amd import scenes
    from envidr_amd.fused import FusedOptions, FusedRenderer
    oc = result.setdefault("other_configs", {})
    # BASELINE configs[1] (no environment MLP; gather / latency-bound regime)
    plain = FusedRenderer.from_scene(scenes.lego_scene(), FusedOptions(dir_sh_degree=4), device=dev)
    pout: dict = {}
    pdt = _time(lambda: plain.render_frame(rays_o, rays_d, None, out=pout, wait=False, image_width=W), 5, dev)
    plain.check_frames()
    psamples = int(plain._frame["last"][1])
    oc["configs[1] hash-grid SDF + diffuse/specular MLPs (SH view dir, no env MLP), 800x800, 1 GPU"] = {
        "rays_per_s": N / pdt, "ms_per_frame": pdt * 1e3, "samples_per_s": psamples / pdt, "samples_per_frame": psamples, "samples_per_ray": psamples / N,
        "roofline": _both_rooflines(psamples, pdt, FLOP_PER_SAMPLE_PLAIN)}
    # the headline frames through the single persistent kernel (envidr_render_rays)
    one = FusedRenderer.from_scene(scenes.toaster_scene(sdf_bias=HEADLINE_SDF_BIAS), device=dev)
    oout: dict = {}
    ocost = torch.zeros(N, dtype=torch.int16, device=dev)
    k = [0]

    def one_frame():
        one.render(rays_o, rays_d, 2 * math.pi * k[0] / 200, extras=True, out=oout, ray_cost=ocost)
        k[0] += 1
    odt = _time(one_frame, 5, dev)
    oc["headline workload as ONE persistent kernel per frame (envidr_render_rays), 800x800, 1 GPU"] = {"rays_per_s": N / odt, "ms_per_frame": odt * 1e3}
    # the relight variant of SURVEY 8d: IDE degree 4, env hidden 160 (shape of the shipped env_net_3.pth), kernel <4,5>
    relight = FusedRenderer.from_scene(scenes.toaster_scene(hidden_env=160, ide_deg=4), FusedOptions(ide_degree=4), device=dev)
    rout: dict = {}
    rdt = _time(lambda: relight.render_frame(rays_o, rays_d, 0.3, out=rout, wait=False, image_width=W), 5, dev)
    relight.check_frames()
    rs = int(relight._frame["last"][1])
    oc["configs[2] relight variant: IDE deg 4 + env MLP 38-160-160-160-12 x2 (shape of the shipped env nets), 800x800, 1 GPU"] = {
        "rays_per_s": N / rdt, "ms_per_frame": rdt * 1e3, "samples_per_frame": rs, "samples_per_ray": rs / N, "roofline": _both_rooflines(rs, rdt, FLOP_PER_SAMPLE_RELIGHT)}
    # BASELINE configs[4] with the geometry cache (NOT the headline): fixed camera, rotating environment
    headline = FusedRenderer.from_scene(scenes.toaster_scene(sdf_bias=HEADLINE_SDF_BIAS), device=dev)
    headline.cache_geometry(rays_o, rays_d, samples_per_ray_hint=HEADLINE_SIZING_HINT)          # (first call: buffer sizing)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    cache = headline.cache_geometry(rays_o, rays_d, samples_per_ray_hint=HEADLINE_SIZING_HINT)
    torch.cuda.synchronize(dev)
    build_ms = (time.perf_counter() - t1) * 1e3
    cout: dict = {}
    cdt = _time(lambda: headline.render_cached(cache, 0.1, out=cout), 5, dev)
    oc["configs[4] env-rotation video of a FIXED camera with the geometry cache (bit-identical frames), 800x800, 1 GPU"] = {
        "rays_per_s": N / cdt, "ms_per_frame": cdt * 1e3, "cache_build_ms": build_ms, "cached_samples": cache.n_samples,
        "samples_per_ray": cache.n_samples / N}
    # split-precision shading mode (env MLP on the fp16 matrix cores with (hi, lo) operand pairs, heads fp32): reported here
    # only, NEVER the headline (whose dtype is f32 throughout); error measured against the fp32 frame of the same view.  Its own
    # roofline is the fp16 MFMA peak: 3 fp16 MFMA products per fp32 product of the environment network.
    del cache
    split_legs = {}
    for tag, fr, hint in (("headline scene", headline, HEADLINE_SIZING_HINT), ("thinner shell of rounds 1-5", FusedRenderer.from_scene(scenes.toaster_scene(), device=dev), 20.0)):
        ref = {k_: v.clone() for k_, v in fr.render_frame(rays_o, rays_d, 0.1, out={}, samples_per_ray_hint=hint).items() if k_ in ("image", "specular_image")}
        sout: dict = {}
        sev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(5)]
        k = [0]

        def split_frame():
            fr.render_frame(rays_o, rays_d, 0.1, out=sout, wait=False, env_precision="f16x2", image_width=W, samples_per_ray_hint=hint,
                            events=sev[k[0] % 5])
            k[0] += 1
        sdt = _time(split_frame, 5, dev)
        fr.check_frames()
        shade_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in sev]))
        geo_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in sev]))
        split = fr.render_frame(rays_o, rays_d, 0.1, out=sout, env_precision="f16x2", samples_per_ray_hint=hint)
        ssamples = int(fr._frame["last"][1])
        srel = {k_: float(torch.linalg.norm(split[k_] - ref[k_]) / torch.linalg.norm(ref[k_])) for k_ in ref}
        f16_flops = 3 * ssamples * 2 * 305_152 / (shade_ms * 1e-3) / 1e12           # three fp16 products per fp32 product of the env MLP, both evaluations
        split_legs[tag] = {"rays_per_s": N / sdt, "ms_per_frame": sdt * 1e3, "geometry_ms": geo_ms, "shading_ms": shade_ms, "samples_per_frame": ssamples,
                           "rel_l2_vs_f32_frame": srel,
                           "roofline": {"bound": "mfma", "achieved": f16_flops, "peak": PEAK_FP16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": f16_flops / PEAK_FP16_MFMA_TFLOPS,
                                        "note": "fp16 MFMA FLOPs actually issued by the environment network (3 per fp32 product: hi*hi + hi*lo + lo*hi) over the "
                                                "HIP-event time of the whole shading stage (environment kernel + fp32 heads), against the dense fp16 MFMA peak"}}
    oc["headline workload in the optional split-precision shading mode (env MLP: fp16 MFMA on (hi, lo) pairs; NOT f32, not comparable to value), 800x800, 1 GPU"] = dict(
        split_legs, dtype="f16x2 pairs (env MLP) + f32 (everything else)", parity_tests="tests/test_split_gpu.py: <= 1e-4 rel-L2 vs the reference's chains and frames")
    del headline
    # A "trained-like" scene (sharp Laplace density, beta = 1e-3; sdf mostly positive in front of the surface): most
    # samples inside the occupancy shell then have alpha = 1 - exp(-sigma dt) == 0 EXACTLY in fp32.  The reference shades them all;
    # the pipeline composites them (they are records) but gathers only the records with a non-zero weight for shading.
    tl_scene = scenes.toaster_scene(beta=1e-3, sdf_bias=0.065)
    tl = {}
    for skip in (True, False):
        fr_tl = FusedRenderer.from_scene(tl_scene, FusedOptions(skip_zero_weight=skip), device=dev)
        tout: dict = {}
        tdt = _time(lambda: fr_tl.render_frame(rays_o, rays_d, 0.2, out=tout, wait=False, image_width=W), 5, dev)
        fr_tl.check_frames()
        tl[skip] = (tdt, int(fr_tl._frame["last"][1]), int(fr_tl._frame["shade_list"][0].item()) if skip else None, tout["image"].clone())
    oc["trained-like scene (beta 1e-3: exact-zero compositing weights), toaster network, 800x800, 1 GPU"] = {
        "rays_per_s": N / tl[True][0], "ms_per_frame": tl[True][0] * 1e3, "samples_per_ray": tl[True][1] / N, "records": tl[True][1],
        "records_shaded": tl[True][2], "ms_per_frame_shading_every_record": tl[False][0] * 1e3,
        "frames_identical": bool(torch.equal(tl[True][3], tl[False][3]))}
    # BASELINE configs[0]'s lineage as the reference itself renders it: the env-sphere mode (configs/neural_renderer.ini -> run_sph:
    # 12 samples around every analytic ray / sphere hit, SDF network 37-64-64-14 with the material parameters concatenated, IDE degree 4,
    # environment MLP 38-160-160-160-12 x2, heads, torch-formula compositing) through NeRFNetwork.render(): the four-launch fused form
    # and the reference-shaped operator chain.  Seeded xavier weights of the shipped shapes (every hit ray has exactly 12 samples: the
    # time does not depend on the weights); parity is tests/test_sph_gpu.py against the imported reference's render.
    from envidr_amd.nerf.network import NeRFNetwork
    from envidr_amd.nerf.options import EnvOptions, neural_renderer_options
    sopt = neural_renderer_options(env_sph_radius=0.95 * 0.8)
    torch.manual_seed(0)
    smodel = NeRFNetwork(encoding="hashgrid", encoding_dir=sopt.encoding_dir, bound=sopt.bound, cuda_ray=False, density_scale=1, min_near=sopt.min_near,
                         density_thresh=sopt.density_thresh, bg_radius=sopt.bg_radius, use_sdf=True, hidden_dim=sopt.hidden_dim, num_layers=sopt.num_layers,
                         num_layers_color=sopt.num_layers_color, hidden_dim_color=sopt.hidden_dim_color, num_levels=sopt.num_levels,
                         geo_feat_dim=sopt.geo_feat_dim, opt=sopt, env_opt=EnvOptions()).to(dev).eval()
    with torch.no_grad():
        smodel.encoder.embeddings.uniform_(-0.1, 0.1)
        smodel.sdf_density.beta.fill_(0.005)
    smat = {"roughness": 0.3, "metallic": 0.2, "color": [20 / 255, 70 / 255, 160 / 255, 1.0]}
    sph = {}
    for res in (400, 800):
        so, sd = (torch.from_numpy(a).to(dev)[None] for a in scenes.camera_rays(res, res, theta=123.0, phi=10.0, radius=4.0, scale=0.8))
        skw = dict(bg_color=1, perturb=False, get_normal_image=False, env_net_index=3, material=smat)
        with torch.no_grad():
            first = smodel.render(so, sd, fused=True, **skw)
            hits = int((first["weights_sum"] > 0).sum().item())
            fdt = _time(lambda: smodel.render(so, sd, fused=True, **skw), 5, dev)
            odt2 = _time(lambda: smodel.render(so, sd, fused=False, **skw), 2, dev)
            other = smodel.render(so, sd, fused=False, **skw)
        sph[f"{res}x{res}"] = {"ms_per_frame": fdt * 1e3, "rays_per_s": res * res / fdt, "hit_rays": hits, "samples": hits * 12,
                               "samples_per_s": hits * 12 / fdt, "operator_chain_ms_per_frame": odt2 * 1e3,
                               "rel_l2_fused_vs_operator_chain": float(torch.linalg.norm(first["image"] - other["image"]) / torch.linalg.norm(other["image"]))}
    oc["configs[0] lineage: env-sphere mode (neural_renderer.ini, run_sph: 12 samples per hit ray, material-conditioned SDF), 1 GPU"] = dict(
        sph, note="BASELINE.md section 2 timed the reference's 1-sample notebook form of this scene on 8 CPU cores: 0.268 s at 400x400, 1.397 s at 800x800; "
                  "run_sph evaluates 12 samples per hit ray and the hash grid + SDF network as well")
    del smodel
    # BASELINE configs[3]: use_renv + indir_ref, three passes per frame through the NeRFRenderer.render() drop-in surface
    from envidr_amd.nerf.options import toaster_options
    iopt = toaster_options(indir_ref=True)
    imodel = NeRFNetwork.from_scene(scenes.toaster_scene(shape=scenes.torus(), seed=3), iopt, device=dev)
    ikw = dict(staged=True, bg_color=1, perturb=False, get_normal_image=True, max_steps=iopt.max_steps, T_thresh=iopt.T_thresh,
               dt_gamma=iopt.dt_gamma, image_width=W)
    idt = _time(lambda: imodel.render(rays_o[None], rays_d[None], **ikw), 3, dev)
    ifr = imodel.fused_renderer()
    ifr.frame_log = {}
    imodel.render(rays_o[None], rays_d[None], **ikw)
    torch.cuda.synchronize(dev)
    ilog = {t: [int(v) for v in x.tolist()] for t, x in ifr.frame_log.items()}
    ifr.frame_log = None
    # algorithmic work of the three passes: geometry (SDF forward + input gradient) on every evaluated sample, full shading on
    # the records of the reflected and main passes, the reflected-radiance MLP + second specular head on the main pass's
    iflop = (sum(v[0] for v in ilog.values()) * FLOP_PER_SAMPLE_GEOMETRY + (ilog.get("indirect-reflected", [0, 0])[1] + ilog.get("indirect-main", [0, 0])[1]) * FLOP_PER_SAMPLE_SHADING
             + ilog.get("indirect-main", [0, 0])[1] * FLOP_PER_SAMPLE_RENV)
    isamples = sum(v[0] for v in ilog.values())
    direct = NeRFNetwork.from_scene(scenes.toaster_scene(shape=scenes.torus(), seed=3), toaster_options(indir_ref=False), device=dev)
    ddt = _time(lambda: direct.render(rays_o[None], rays_d[None], **ikw), 3, dev)
    oc["configs[3] toaster network + use_renv + indir_ref (3 passes per frame), torus scene, 800x800, 1 GPU"] = {
        "primary_rays_per_s": N / idt, "ms_per_frame": idt * 1e3, "direct_frame_of_the_same_scene_ms": ddt * 1e3, "ratio_to_direct": idt / ddt,
        "samples_per_primary_ray": ilog.get("indirect-geometry", [0, 0])[1] / N,
        "passes": "geometry of the primary rays (march + SDF network) -> reflected rays (a full frame of their own) -> main pass = shading + "
                  "composite of the FIRST pass's records with the reflected radiance (same rays, same samples: not marched or evaluated again)",
        "samples_evaluated_and_records_per_pass": {t: v[:2] for t, v in ilog.items()},
        "roofline": {"hbm": {"achieved": isamples * HASH_BYTES_PER_SAMPLE / idt / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                             "frac": isamples * HASH_BYTES_PER_SAMPLE / idt / 1e9 / PEAK_HBM_GBPS},
                     "mfma": {"achieved": iflop / idt / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                              "frac": iflop / idt / 1e12 / PEAK_FP32_MFMA_TFLOPS}}}


def training_and_loop_legs(result) -> None:
    """The widening rows next to the frame pipeline (INTEGRATION.md way 1), as context: one training step of the toaster network
    (tools/train_step_bench.py) and the reference-shaped operator loop on the headline frame (tools/loop_frame_bench.py).  Each runs in its
    own process after the timed region (its failure or time-out is recorded, never the bench's); neither is `value`."""
    import re
    import subprocess
    oc = result.setdefault("other_configs", {})
    for key, script, argv in (("training step (run_cuda's training branch: 4 096 rays, ~144 k samples, eikonal loss, Adam), 1 GPU", "train_step_bench.py", ["20"]),
                              ("reference-shaped operator loop (fused=False) on the headline frame, 800x800, 1 GPU", "loop_frame_bench.py", ["3"])):
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), *argv], capture_output=True, text=True, timeout=240, cwd=ROOT)
            lines = [l for l in r.stdout.splitlines() if " ms per " in l]
            if r.returncode != 0 or not lines:
                oc[key] = {"error": (r.stderr or r.stdout)[-300:]}
                continue
            entry = {"lines": lines}
            for l in lines:
                m = re.search(r"([0-9.]+) ms per (step|800x800 frame)", l)
                if not m:
                    continue
                if l.startswith("steady state"):
                    entry["ms_per_step_steady_state"] = float(m.group(1))
                elif l.startswith("training step"):
                    entry["ms_per_step_first_16_steps_regime"] = float(m.group(1))
                elif l.startswith("fused=False"):
                    entry["ms_per_frame"] = float(m.group(1))
            oc[key] = entry
        except Exception as e:      # noqa: BLE001
            oc[key] = {"error": repr(e)[:300]}


def main() -> None:
    args = parse(sys.argv[1:])
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's command line (`python bench.py --gpus N ...`) works by itself: start the N ranks here
        import torch.multiprocessing as mp
        if not args.stub and torch.cuda.device_count() < args.gpus:
            _preflight_failed({"ok": False, "stage": "device_count", "rank": 0, "world": args.gpus,
                               "error": f"torch.cuda.device_count() = {torch.cuda.device_count()} < --gpus {args.gpus}"})
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        mp.spawn(_spawned, args=(args.gpus, _free_port(), sys.argv[1:]), nprocs=args.gpus, join=True)
        return
    run(sys.argv[1:])


if __name__ == "__main__":
    main()
