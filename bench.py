#!/usr/bin/env python
"""Benchmark of the ENVIDR render hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric (BASELINE.json): rendered rays/s at 800x800, max 1024 samples/ray, plus PSNR of the GPU image
against the CPU reference restatement on a sample of the same scene.

One "step" = every rank renders ONE full 800x800 view (640 000 primary rays) of the synthetic
toaster scene with the fused persistent kernel -- march + hash grid + SDF MLP + analytic normals +
2x IDE + 2x environment MLP + diffuse/specular heads + compositing, all four auxiliary images on
(normal / diffuse / specular / roughness, as the reference's toaster.ini renders them) -- and the
finished RGB frame is gathered to rank 0 (RCCL gather over xGMI; no-op at N = 1).  Views are the
env-rotation video frames of BASELINE config #5: view v = step * N + rank gets env rotation
2 pi v / 200, so per-GPU work is fixed as N grows (weak scaling).  Inputs (rays, table, weights) are
resident in HBM before the timed region.

Printed by rank 0 as ONE JSON line; see DESIGN.md section "Measurement" for the roofline arithmetic.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 800
FLOP_PER_SAMPLE = 650_880          # dense-layer FLOPs per shaded sample, toaster network
FLOP_PER_SAMPLE_SHADING = 624_192  # of which in the shading pass: env MLP 305 152 x 2 + diffuse 1 728 + specular 12 160
                                   # (the other 26 688 -- SDF network forward + input gradient -- run in the geometry pass)
HASH_BYTES_PER_SAMPLE = 1024       # 16 levels x 8 corners x 8 B gathered per sample
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32-input MFMA dense peak
CPU_SAMPLE_RES = 400               # cpu_baseline renders a 400x400 frame of the same scene/camera
CPU_THREAD_CANDIDATES = (16, 32, 64)


def cpu_baseline(scene, env_rot: float) -> dict:
    """the CPU restatement (oracle: C/OpenMP ops + torch CPU GEMMs, reference n_step schedule) timed
    on the host cores on a bounded sample of the same workload"""
    from envidr_amd import scenes
    from oracle.py import render_oracle as ro
    opt = ro.RenderOptions(ide_mode="torch")
    # thread count: all hardware threads is NOT the fastest for these small GEMMs (on the 2x64-core
    # GPU host 256 threads run ~400x slower than 16); pick the best of a few candidates on a small probe
    probe_o, probe_d = scenes.camera_rays(96, 96)
    best, cores = None, 1
    for th in sorted({min(c, os.cpu_count() or 1) for c in CPU_THREAD_CANDIDATES}):
        torch.set_num_threads(th)
        ro.render_rays(scene, probe_o[:256], probe_d[:256], opt, env_rot)    # warm-up (library loads, thread pools)
        t0 = time.perf_counter()
        ro.render_rays(scene, probe_o, probe_d, opt, env_rot)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, cores = dt, th
    torch.set_num_threads(cores)
    rays_o, rays_d = scenes.camera_rays(CPU_SAMPLE_RES, CPU_SAMPLE_RES)
    t0 = time.perf_counter()
    res = ro.render_rays(scene, rays_o, rays_d, opt, env_rot)
    dt = time.perf_counter() - t0
    n = CPU_SAMPLE_RES * CPU_SAMPLE_RES
    return {"value": n / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{CPU_SAMPLE_RES}x{CPU_SAMPLE_RES} frame of the same scene and camera ({n} rays, {res['n_samples']} samples, "
                      f"{dt:.1f} s): oracle/ C+OpenMP ops + torch CPU fp32 GEMMs, reference n_step schedule, best of "
                      f"{list(CPU_THREAD_CANDIDATES)} threads on a {os.cpu_count()}-thread host",
            "samples_per_s": res["n_samples"] / dt, "image": res["image"]}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (profiling runs)")
    ap.add_argument("--path", choices=["two-phase", "fused"], default="two-phase",
                    help="two-phase: geometry pass + shading pass per frame (default, faster); fused: one persistent kernel per frame")
    ap.add_argument("--headline-only", action="store_true", help="skip the CPU leg and the other_configs renders, so that a "
                    "profiler sees only the headline kernel's launches")
    args = ap.parse_args()

    from envidr_amd import parallel, scenes
    from envidr_amd.fused import FusedRenderer
    import torch.distributed as dist

    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    scene = scenes.toaster_scene()
    renderer = FusedRenderer.from_scene(scene, device=dev)
    rays_o, rays_d = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(H, W))
    N = H * W
    out: dict = {}
    # scheduling hint (DESIGN.md "work-list order"): samples per ray of the previous frame of this camera; orders the
    # persistent kernel's work list longest ray first.  Order only: every frame still marches and shades every sample.
    ray_cost = torch.zeros(N, dtype=torch.int16, device=dev)
    gather_list = [torch.empty(N, 3, device=dev) for _ in range(world)] if (world > 1 and rank == 0) else None

    def env_rot(view: int) -> float:
        return 2 * math.pi * (view % 200) / 200

    two_phase = args.path == "two-phase"

    def frame(view: int, events=None):
        if two_phase:
            return renderer.render_two_phase(rays_o, rays_d, env_rot(view), out=out, ray_cost=ray_cost, events=events)
        if events:
            events[0].record()
        res = renderer.render(rays_o, rays_d, env_rot(view), extras=True, stats=True, out=out, ray_cost=ray_cost)
        if events:
            events[1].record()
        return res

    def step(i: int) -> None:
        res = frame(i * world + rank)
        if world > 1:
            dist.gather(res["image"], gather_list=gather_list, dst=0)

    for i in range(args.warmup):
        step(i)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # HIP events on the stream the kernels are launched on (torch's current stream), recorded at the pass boundaries
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    samples = 0
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        res = frame((args.warmup + i) * world + rank, ev[i])
        if world > 1:
            dist.gather(res["image"], gather_list=gather_list, dst=0)
    fence()
    dt = time.perf_counter() - t0
    if two_phase:
        geometry_ms, kernel_ms, composite_ms = (float(np.mean([e[j].elapsed_time(e[j + 1]) for e in ev])) for j in range(3))
        samples = int(res["n_records"])        # samples composited = records shaded in the last frame
    else:
        kernel_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        samples = int(out["stats"][0].item())  # samples shaded in the last frame
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        rays_per_s = world * N * args.steps / dt
        flops = samples * (FLOP_PER_SAMPLE_SHADING if two_phase else FLOP_PER_SAMPLE) / (kernel_ms * 1e-3) / 1e12
        result = {
            "metric": "rendered rays/s at 800x800, 1024 max samples/ray", "value": rays_per_s, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]/[4] network (toaster.ini: hash L16xC2 + SDF 32-64-64-15 + IDE deg5 + env MLP "
                                   "72-256-256-256-12 x2 + diffuse/specular heads) on a synthetic shell scene, 800x800 view per GPU per "
                                   "step, env-rotation video frames sharded by view, normal/diffuse/specular/roughness images on",
                       "rays_per_step_per_gpu": N, "samples_per_frame": samples, "samples_per_ray": samples / N,
                       "max_steps": 1024, "T_thresh": 1e-4, "parallelism": f"views x{world} + RCCL image gather",
                       "schedule": ("two-phase frame: geometry pass (march + hash grid + SDF network + normals, one record per "
                                    "composited sample) -> shading pass (k_shade_samples) -> per-ray composite; every frame from scratch"
                                    if two_phase else "one persistent kernel per frame")},
            "samples_per_s": world * samples * args.steps / dt,
            "roofline": {"bound": "mfma", "achieved": flops, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                         "kernel": "k_shade_samples<5,8>" if two_phase else "k_render_persistent<5,8,0>", "kernel_ms": kernel_ms,
                         "kernel_ms_note": ("HIP events around the shading launch (envidr_shade_records) on its stream" if two_phase else
                                            "HIP events around one envidr_render_rays call on its stream: the persistent kernel plus "
                                            "its two pre-pass kernels (k_first_hit + k_order_hits, about 0.5 ms)"),
                         "algorithmic_flop_per_sample": FLOP_PER_SAMPLE_SHADING if two_phase else FLOP_PER_SAMPLE,
                         "samples_per_launch": samples,
                         "hbm_view": {"bound": "hbm", "achieved": samples * HASH_BYTES_PER_SAMPLE / (kernel_ms * 1e-3) / 1e9,
                                      "peak": 8000.0, "unit": "GB/s"}},
        }
        if two_phase:
            result["frame"] = {"geometry_ms": geometry_ms, "shading_ms": kernel_ms, "composite_ms": composite_ms,
                               "whole_frame_mfma_TFLOPs": samples * FLOP_PER_SAMPLE / (dt / args.steps) / 1e12,
                               "whole_frame_mfma_frac": samples * FLOP_PER_SAMPLE / (dt / args.steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                               "hbm_record_bytes_per_sample": 80}
        if world == 1 and not args.headline_only:
            other_configs(result, scenes, FusedRenderer, rays_o, rays_d, dev, N)
        prof = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(prof):
            try:
                result["roofline"]["traffic"] = json.load(open(prof)).get("hbm_bytes_per_launch")
            except Exception:
                pass
        if not args.no_cpu_baseline and not args.headline_only and world == 1:
            cpu = cpu_baseline(scene, env_rot(0))
            # PSNR of the GPU render vs the CPU reference restatement on the same sample
            so, sd = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(CPU_SAMPLE_RES, CPU_SAMPLE_RES))
            g = renderer.render(so, sd, env_rot(0), extras=False)["image"].cpu().numpy()
            ref = cpu.pop("image")
            mse = float(np.mean((g.astype(np.float64) - ref) ** 2))
            result["psnr_vs_cpu_reference_db"] = -10 * math.log10(max(mse, 1e-20))
            result["rel_l2_vs_cpu_reference"] = float(np.linalg.norm(g.astype(np.float64) - ref) / np.linalg.norm(ref))
            result["cpu_baseline"] = cpu
            result["speedup_vs_cpu_baseline"] = rays_per_s / cpu["value"]
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def other_configs(result, scenes, FusedRenderer, rays_o, rays_d, dev, N) -> None:
    """the other single-GPU BASELINE configurations on the same camera (parity-test configurations, not the headline)"""
    # BASELINE configs[1] (no environment MLP; gather/latency-bound regime)
    from envidr_amd.fused import FusedOptions
    plain = FusedRenderer.from_scene(scenes.lego_scene(), FusedOptions(dir_sh_degree=4), device=dev)
    pout: dict = {}
    plain.render(rays_o, rays_d, None, extras=True, stats=True, out=pout)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for _ in range(5):
        plain.render(rays_o, rays_d, None, extras=True, stats=True, out=pout)
    torch.cuda.synchronize(dev)
    pdt = (time.perf_counter() - t1) / 5
    psamples = int(pout["stats"][0].item())
    result["other_configs"] = {"configs[1] hash-grid SDF + diffuse/specular MLPs (SH view dir, no env MLP), 800x800, 1 GPU": {
        "rays_per_s": N / pdt, "ms_per_frame": pdt * 1e3, "samples_per_s": psamples / pdt,
        "hbm_algorithmic_GBps": psamples * HASH_BYTES_PER_SAMPLE / pdt / 1e9,
        "mfma_algorithmic_TFLOPs": psamples * 41_984 / pdt / 1e12}}
    # the same headline frames through the single persistent kernel (the default of the drop-in renderer)
    one = FusedRenderer.from_scene(scenes.toaster_scene(), device=dev)
    oout: dict = {}
    ocost = torch.zeros(N, dtype=torch.int16, device=dev)
    for i in range(2):
        one.render(rays_o, rays_d, 0.1 * i, extras=True, out=oout, ray_cost=ocost)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for i in range(5):
        one.render(rays_o, rays_d, 2 * math.pi * i / 200, extras=True, out=oout, ray_cost=ocost)
    torch.cuda.synchronize(dev)
    odt = (time.perf_counter() - t1) / 5
    result.setdefault("other_configs", {})["headline workload as ONE persistent kernel per frame (envidr_render_rays), 800x800, 1 GPU"] = {
        "rays_per_s": N / odt, "ms_per_frame": odt * 1e3}
    # BASELINE configs[4] with the geometry cache (NOT the headline, where every frame marches and shades
    # from scratch): fixed camera, rotating environment: geometry once, then shading + compositing per frame
    headline = FusedRenderer.from_scene(scenes.toaster_scene(), device=dev)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    cache = headline.cache_geometry(rays_o, rays_d)
    torch.cuda.synchronize(dev)
    build_ms = (time.perf_counter() - t1) * 1e3
    cout: dict = {}
    headline.render_cached(cache, 0.1, out=cout)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for i in range(5):
        headline.render_cached(cache, 2 * math.pi * i / 200, out=cout)
    torch.cuda.synchronize(dev)
    cdt = (time.perf_counter() - t1) / 5
    result["other_configs"]["configs[4] env-rotation video of a FIXED camera with the geometry cache (bit-identical frames), 800x800, 1 GPU"] = {
        "rays_per_s": N / cdt, "ms_per_frame": cdt * 1e3, "cache_build_ms": build_ms, "cached_samples": cache.n_samples}
    # BASELINE configs[3]: use_renv + indir_ref, three fused passes per frame (geometry -> reflected rays -> main pass
    # with reflected radiance) through the NeRFRenderer.render() drop-in surface, concave (torus) scene
    from envidr_amd.nerf.network import NeRFNetwork
    from envidr_amd.nerf.options import toaster_options
    iopt = toaster_options(indir_ref=True)
    imodel = NeRFNetwork.from_scene(scenes.toaster_scene(shape=scenes.torus(), seed=3), iopt, device=dev)
    ikw = dict(staged=True, bg_color=1, perturb=False, get_normal_image=True, max_steps=iopt.max_steps, T_thresh=iopt.T_thresh,
               dt_gamma=iopt.dt_gamma)
    imodel.render(rays_o[None], rays_d[None], **ikw)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for _ in range(3):
        imodel.render(rays_o[None], rays_d[None], **ikw)
    torch.cuda.synchronize(dev)
    idt = (time.perf_counter() - t1) / 3
    result["other_configs"]["configs[3] toaster network + use_renv + indir_ref (3 passes per frame), torus scene, 800x800, 1 GPU"] = {
        "primary_rays_per_s": N / idt, "ms_per_frame": idt * 1e3}


if __name__ == "__main__":
    main()
