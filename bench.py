#!/usr/bin/env python3
"""bench.py — the reference's headline workload on N MI355X of one node.

Metric (BASELINE.json): Mrays/s, whole job, final random-spheres scene at 1200 x 800, 500 spp, depth 50
(configs[1]); a "ray" is one ray segment = one call of HittableList::hit (main.cpp:90), counted by the
kernel itself.  A "step" renders one frame of that configuration: every rank renders its interleaved
8 x 8-pixel tiles (scene and camera already resident in HBM), the tile buffers are gathered with one
all_gather over RCCL and de-interleaved to scanline order on every rank.  Total work is fixed as N grows
(strong scaling: the image is sharded, BASELINE.json north_star).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Besides the contract's fields it carries
  roofline     : the render kernel against the FP32 vector peak, algorithmic flops = 17 per ray-sphere
                 test x 484 spheres per ray, kernel time from HIP events on the kernel's stream
  cpu_baseline : the reference compiled unmodified (oracle/_ref/rtiow_ref, kind "reference") or, if that
                 binary is absent, the oracle restatement (kind "port"), one thread, on BASELINE.json configs[0]
                 in full (400x225, 10 spp, depth 50: ~20 s; its rays and output hash are pinned).
"""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import raytracinginoneweekend_amd as rt  # noqa: E402
from raytracinginoneweekend_amd.distributed import ShardedRenderer  # noqa: E402

W, H, SPP, DEPTH = 1200, 800, 500, 50
FLOP_PER_TEST = 17          # oc 3, half_b 5, |oc|^2 5, -r^2 1, disc 3
FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md "Peak FP32 (vector)"
# Scenes whose filter runs on the matrix pipe: a word of 32 spheres x 64 rays = 2048 tests costs a SIMD 2 x 32 cycles of
# v_mfma_f32_32x32x16_f16 AND 2 fp32 FMAs per test and lane = 128 cycles of VALU issue.  The two pipes of a SIMD run side by
# side (SQ_VALU_MFMA_COEXEC_CYCLES of this kernel: 0.67 of its matrix cycles have a vector instruction executing too,
# profiles/r03_summary.md; MI355X_MICROARCH.md: separate pipes), so the mapping's bound is the LONGER of the two, the 128 VALU
# cycles: 2048 x 17 algorithmic flop / 128 cycles x 1024 SIMDs x 2.4 GHz = 668.5 TFLOP/s.  (Round 2 priced the mapping at
# 64 + 128 = 192 cycles, 445.6 TFLOP/s, on the strength of a microbenchmark that the counter contradicts.)  The fp32 vector peak
# alone is no upper bound of the mapping: most of a test's arithmetic is f16 products on the matrix pipe.
ISSUE_BOUND_PEAK_TFLOPS = 2048 * 17 / 128 * 1024 * 2.4e9 / 1e12
KERNEL_TAG = "r03-a (dot-product filter: squared distance minus r^2 of 32 spheres x 32 rays as one f16 MFMA from two-part splits of the centre's monomials and the ray's coefficients, the MFMAs of a word issued back to back as one asm statement, threshold and bit accumulation in 2 fp32 FMAs per test; candidate mask words in LDS; per-sample fixed-point sums, run schedule sized per rank)"   # bump when the kernel changes


# BASELINE.json configs[0] — the reference's own CPU-runnable case: `400 225 10 50`.  The reference is deterministic
# (default-seeded mt19937 statics), so this run always traces the same rays and writes the same file: both are pinned
# (tests/test_oracle_golden.py checks the oracle reproduces them).
C1_ARGV = ["400", "225", "10", "50"]
C1_RAYS = 2604684
C1_PPM_SHA256 = "f87897196e761b05544c5dcd940b34128661b94e2ff10402ac4859aae7e31c34"


def cpu_baseline():
    """Single-thread CPU time of the reference on BASELINE.json configs[0] (same scene, camera and depth as the GPU
    workload; 0.9 M samples, ~20 s): the compiled reference itself, else the oracle restatement."""
    import hashlib
    from oracle import pyoracle as po   # the only place bench.py touches oracle/: the baseline leg
    ref_bin = po.REF_BINARY_PATH
    cores = 1
    W1, H1, spp1 = int(C1_ARGV[0]), int(C1_ARGV[1]), int(C1_ARGV[2])
    try:
        if os.path.exists(ref_bin) and os.access(ref_bin, os.X_OK):
            # its own ProfileLogScope times the "Image Generation" scope (main.cpp:63)
            with tempfile.TemporaryDirectory() as td:
                r = subprocess.run([ref_bin, *C1_ARGV], cwd=td, capture_output=True, text=True, timeout=900)
                sha = hashlib.sha256(open(os.path.join(td, "image_binary.ppm"), "rb").read()).hexdigest()
            m = re.search(r"Image Generation took ([0-9.eE+-]+) seconds", r.stderr)
            secs = float(m.group(1))
            if sha != C1_PPM_SHA256:
                raise RuntimeError("the reference binary wrote an image with an unexpected hash " + sha)
            return {"value": C1_RAYS / secs / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "reference",
                    "sample": "BASELINE.json configs[0] in full: reference executable (g++ -O2 -ffp-contract=off, unmodified sources), final scene, "
                              "400x225, 10 spp, depth 50: %d rays in %.2f s (its own 'Image Generation' scope; output PPM hash = golden)" % (C1_RAYS, secs),
                    "samples_per_s": W1 * H1 * spp1 / secs}
    except Exception as e:   # fall through to the port
        print("cpu_baseline: reference binary failed (%s); timing the oracle port instead" % e, file=sys.stderr)
    o = po.Oracle()
    o.random_scene()
    t0 = time.perf_counter()
    _, u8, seg = o.render_rows(po.stock_camera(po.cli_dims(C1_ARGV)[4]), W1, H1, spp1, int(C1_ARGV[3]))
    secs = time.perf_counter() - t0
    return {"value": seg / secs / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "BASELINE.json configs[0] in full: oracle/cpu_ref.cpp (-O2 -ffp-contract=off), 400x225, 10 spp, depth 50: %d rays in %.2f s" % (seg, secs),
            "samples_per_s": W1 * H1 * spp1 / secs}


def workload_labels(args, n_spheres):
    """metric / workload strings and whether this run is the headline workload (BASELINE.json configs[1] exactly: stock scene,
    list walk, 1200 x 800, 500 spp, depth 50).  Any other size or scene names its own BASELINE config (or says it has none)."""
    size = (args.width, args.height, args.spp, args.depth)
    stock = args.scene_k <= 0 and not args.glass and args.mixed <= 0
    headline = stock and args.accel == "list" and size == (W, H, SPP, DEPTH)
    walk = "uniform grid (bit-identical image, fewer sphere tests; not the roofline's algorithm)" if args.accel == "grid" else "list walk"
    dims = f"{args.width}x{args.height}, {args.spp} spp, depth {args.depth}"
    if args.mixed > 0:
        cfg, scene = "no BASELINE config (user scene)", f"{n_spheres} spheres of varied radius (general words)"
    elif args.glass:
        cfg, scene = "BASELINE.json configs[4]", f"glass/metal-heavy scene ({n_spheres} spheres, material thresholds 0.1/0.5)"
    elif args.scene_k > 0:
        cfg, scene = "BASELINE.json configs[3] (scaled list)", f"{n_spheres} spheres on [-{args.scene_k},{args.scene_k})^2"
    else:
        scene = f"book-1 final random-spheres scene ({n_spheres} spheres, libstdc++ default-seeded)"
        cfg = ("BASELINE.json configs[1]" if size == (W, H, SPP, DEPTH) else "BASELINE.json configs[2] (on %d GPU(s))" % args.gpus if size == (3840, 2160, 1000, 50) else
               "BASELINE.json configs[0] (the reference's CPU case, here on the GPU)" if size == (400, 225, 10, 50) else "no BASELINE config (stock scene at another size)")
    metric = ("Mrays/s (whole node), final scene 1200x800 500spp" if headline else
              f"Mrays/s (whole node), NOT the headline workload: {cfg}: {scene}, {dims}, {'uniform grid' if args.accel == 'grid' else 'list walk'}")
    return {"headline": headline, "metric": metric, "workload": f"{cfg}: {scene}, {dims}, stock camera, closest hit by {walk}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=W)
    ap.add_argument("--height", type=int, default=H)
    ap.add_argument("--spp", type=int, default=SPP)
    ap.add_argument("--depth", type=int, default=DEPTH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=1, choices=(1, 2),
                    help="2: consecutive frames alternate between two contexts, each rendering on its own stream, so that the tail of one frame's "
                         "persistent kernel overlaps the start of the next; gathers stay in order on one stream (throughput experiment, "
                         "DESIGN.md §7; per-kernel event times then include the overlap)")
    ap.add_argument("--scene-k", type=int, default=0,
                    help="0: the stock 484-sphere scene (the metric's workload).  K > 0: BASELINE.json config 4, the same generator on a [-K,K)^2 grid (K=50: 10k spheres)")
    ap.add_argument("--glass", action="store_true",
                    help="BASELINE.json configs[4]: the stock generator with the material thresholds of main.cpp:130,133 moved from 0.8/0.95 to 0.1/0.5 "
                         "(10 %% diffuse, 40 %% metal, 50 %% glass); use with --depth 100")
    ap.add_argument("--mixed", type=int, default=0,
                    help="N > 0: a user scene of N spheres of varied radius (general words of the matrix-pipe layout; not a BASELINE config)")
    ap.add_argument("--accel", choices=("list", "grid"), default="list",
                    help="list: the reference's list walk (the metric's path, roofline applies).  grid: uniform grid (SURVEY.md §8f-3), same image, reported separately")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    ctx = rt.Context(local_rank)
    if sum(bool(x) for x in (args.scene_k > 0, args.glass, args.mixed > 0)) > 1:
        raise SystemExit("--scene-k, --glass and --mixed exclude each other")
    scene = (rt.mixed_radii_scene(args.mixed) if args.mixed > 0 else rt.scaled_scene(diffuse_below=0.1, metal_below=0.5) if args.glass else
             rt.stock_scene() if args.scene_k <= 0 else rt.scaled_scene(k=args.scene_k))
    ctx.set_scene(scene)
    if args.accel == "grid" and ctx.set_accel(rt.ACCEL_GRID) != rt.ACCEL_GRID:
        raise SystemExit("the grid was refused for this scene: " + rt.lib().rtk_last_error().decode())
    cam = rt.stock_camera(float(np.float32(args.width) / np.float32(args.height)))
    ctxs = [ctx]
    for _ in range(1, args.frames_in_flight):
        c2 = rt.Context(local_rank)
        c2.set_scene(scene)
        if args.accel == "grid":
            c2.set_accel(rt.ACCEL_GRID)
        ctxs.append(c2)
    # one renderer: with several contexts it alternates them frame by frame, gathers stay on one stream in order
    sr = ShardedRenderer(ctx, args.width, args.height, rank, world, device, extra_contexts=ctxs[1:])

    def step(i):
        sr.render(cam, args.spp, args.depth, seed=i)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(1000 + i)
    fence()
    for c in ctxs:
        c.reset_stats()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    st = ctx.get_stats()
    for c in ctxs[1:]:                       # counters of the other frames in flight
        o = c.get_stats()
        for key in ("rays", "samples", "sphere_tests", "exact_tests", "lane_slots", "kernel_ms", "frames", "frames_timed"):
            st[key] += o[key]

    # the kernel's drain share is a per-launch figure (min / max over the waves of everything since the last reset):
    # one more frame, untimed, on cleared counters
    for c in ctxs:
        c.reset_stats()
    step(args.steps)
    fence()
    drain_frac = ctx.get_stats().get("drain_frac")

    per_rank_ms = [st["kernel_ms"] / max(1, st["frames_timed"])]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank_ms[0])
        per_rank_ms = [float(x) for x in gathered]
    tot = torch.tensor([float(st["rays"]), float(st["samples"]), float(st["kernel_ms"]), float(st["frames_timed"]), float(st["lane_slots"]),
                        float(st["exact_tests"])], dtype=torch.float64, device=device)
    tmax = torch.tensor([elapsed, st["kernel_ms"] / max(1, st["frames_timed"])], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    rays, samples, _, _, lane_slots, exact = (float(x) for x in tot.tolist())
    elapsed_max, kernel_ms_max = (float(x) for x in tmax.tolist())

    if rank == 0:
        n_spheres = len(scene["cx"])
        mrays = rays / elapsed_max / 1e6
        # roofline of the dominant kernel (render_kernel), on rank 0's own launches
        k_rays = st["rays"] / max(1, st["frames"])                       # rays one launch traces on this rank
        k_ms = st["kernel_ms"] / max(1, st["frames_timed"])
        achieved = k_rays * n_spheres * FLOP_PER_TEST / (k_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # only a measurement of THIS kernel build counts (the tag is bumped whenever the kernel's memory behaviour changes)
                traffic = tj.get("bytes_per_launch") if tj.get("kernel_tag") == KERNEL_TAG else None      # measured on configs[1] at one GPU: see below
            except Exception:
                traffic = None
        wl = workload_labels(args, n_spheres)
        headline = wl["headline"]
        counters = {}
        if not (headline and world == 1):
            traffic = None          # the counters were collected on configs[1] at one GPU only
        elif os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("kernel_tag") == KERNEL_TAG:
                    counters = {k: tj[k] for k in ("valu_issue_util", "mfma_busy", "mfma_valu_coexec", "source") if k in tj}
            except Exception:
                counters = {}
        lay = ctx.scene_layout()
        n_class = 32 * lay["class_words"]
        # the filter runs on the matrix pipe for LDS-resident scenes and for streamed scenes whose words have tiles in their own
        # frames (rtk_scene_layout: class words > 0); the staged vector-ALU walk is priced against the fp32 vector peak
        on_matrix_pipe = n_spheres <= 1024 or lay["class_words"] > 0
        peak = ISSUE_BOUND_PEAK_TFLOPS if on_matrix_pipe else FP32_VECTOR_PEAK_TFLOPS
        filt_valu = 2.0 if on_matrix_pipe else 8.0                                      # vector instructions per ray-sphere filter test
        alg_bytes = args.width * args.height * 3 * (8 + 4) / world                       # per launch: one u64 accumulator + one fp32 sum per channel
        out = {
            "metric": wl["metric"],
            "value": mrays, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["workload"],
                       "sharding": f"8x8-pixel tiles interleaved over {world} rank(s), one all_gather of fp32 tile sums per frame",
                       "rays_per_sample": rays / samples, "samples_per_s": samples / elapsed_max, "frames_in_flight": args.frames_in_flight},
            "roofline": {"bound": "simd_issue (mfma f16 + valu fp32)" if on_matrix_pipe else "valu_fp32", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "fp32_vector_peak": FP32_VECTOR_PEAK_TFLOPS, "frac_of_fp32_vector_peak": achieved / FP32_VECTOR_PEAK_TFLOPS,
                         "traffic": traffic, "kernel": "render_kernel<true>" if n_spheres <= 1024 else "render_kernel<false>", "kernel_tag": KERNEL_TAG, "kernel_ms": k_ms, "kernel_ms_max_over_ranks": kernel_ms_max,
                         "flop_per_ray": FLOP_PER_TEST * n_spheres, "rays_per_launch": k_rays, "per_rank_kernel_ms": per_rank_ms,
                         "basis": "ALGORITHMIC flops (17 per ray-sphere test x spheres x rays, SURVEY.md §8d), not executed ones, " +
                                  ("against the issue bound of the filter's mapping (per 2048 tests 64 cycles of MFMA and 128 of VALU per SIMD, which run side by side — measured, "
                                   "mfma_valu_coexec — so the bound is the 128: 668.5 TFLOP/s; round 2 divided by 445.6 = the two added; "
                                   "frac_of_fp32_vector_peak is the measure of rounds 1-2, no longer an upper bound): " if on_matrix_pipe else "against the fp32 VECTOR peak: ") +
                                  "the kernel runs a conservative "
                                  "filter per test (" + ("one dot product of split-f16 monomials and coefficients on the matrix pipe, 2 x v_mfma_f32_32x32x16_f16 per class word of 32 spheres x 64 rays, then 2 fp32 vector FMAs"
                                                         + ("" if n_spheres <= 1024 else "; streamed scene: per word the ray's linear coefficients and constant are rebuilt in the word's own frame, ~45 vector instructions")
                                                         if on_matrix_pipe else "8 fp32 vector instructions") +
                                  ") plus ~2 exact reference tests per ray; executed-instruction counts from counters: profiles/*_summary.md",
                         "filter_on_matrix_pipe": on_matrix_pipe,
                         "filter_valu_per_test": filt_valu, "class_words": lay["class_words"], "lds_bytes_per_workgroup": lay["lds_bytes_per_workgroup"],
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "lane_utilisation": rays / max(1.0, lane_slots), "exact_tests_per_ray": exact / max(1.0, rays),
                         "drain_frac_last_frame": drain_frac,
                         # hardware counters of this kernel build on this workload (rocprofv3 --pmc, profiles/<round>_summary.md; replayed by
                         # kernel tag, not collected in this run: counters need the profiler): fractions of the SIMDs' cycles
                         "counters": counters or None},
        }
        if args.accel == "grid":
            # the 17-flop-per-test model prices the list walk; with a grid the useful figure is tests per ray
            out["roofline"] = {"bound": "valu_fp32", "achieved": None, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
                               "kernel": "render_kernel<false, true>", "kernel_ms": k_ms, "exact_tests_per_ray": exact / max(1.0, rays),
                               "note": "uniform grid: algorithmic work differs from the list walk; compare Mrays/s with --accel list on the same scene"}
        if world == 1 and not args.no_cpu_baseline and headline:
            out["cpu_baseline"] = cpu_baseline()
            out["vs_cpu_baseline"] = mrays / out["cpu_baseline"]["value"]     # vs_baseline stays null: the reference publishes no number
        print(json.dumps(out), flush=True)
    # tear down in order: the context must not outlive the torch stream it was pointed at
    torch.cuda.synchronize()
    for c in ctxs:
        c.set_stream(0)
        c.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
