"""GPU study of the sample-run schedule (rtk_params.run_len): kernel time, lane utilisation and number of accumulator
flushes per pixel for rank 0 of `world` ranks of the 1200x800x500 frame.  run_len > 0: fixed runs; 0: shipped default;
< 0: decreasing runs L = ceil(remaining / k) with k = -run_len.   usage: python tools/sched_sweep.py [world ...]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracinginoneweekend_amd as rt

W, H, SPP, DEPTH = 1200, 800, 500, 50
worlds = [int(a) for a in sys.argv[1:]] or [1, 8]
ctx = rt.Context(0)
ctx.set_scene(rt.stock_scene())
cam = rt.stock_camera(W / H)
for world in worlds:
    nlt = rt.lib().rtk_local_tile_count(W, H, 8, 8, 0, world)
    out = torch.zeros((nlt, 64, 3), dtype=torch.float32, device="cuda:0")
    for run_len in (0, -256, -64, -32, 1):
        # the partition is sized for THIS rank's pixels (render_tiles_impl: n_local_px = local tiles x 64), not for the full frame
        n_runs = rt.lib().rtk_run_schedule(SPP, 64, int(nlt), run_len, None, 0)
        ms = []
        for rep in range(3):
            p = rt.make_params(W, H, SPP, DEPTH, rep, 0.001, 0, world, 8, 8, run_len, 0)
            st = ctx.render_tiles_device(cam, p, out.data_ptr(), want_stats=True)
            ms.append(st["kernel_ms"])
        mean_passes = st["lane_slots"] / 64 / st["waves"]
        print("world %d run_len %4d: %3d runs per pixel, kernel ms %s, lane utilisation %.4f, passes per wave mean %.0f max %d, drain %.3f, Grays/s (x world) %.2f"
              % (world, run_len, n_runs, ["%.2f" % m for m in ms], st["rays"] / st["lane_slots"], mean_passes, st["max_wave_passes"],
                 st["drain_frac"], st["rays"] / min(ms[1:]) / 1e6 * world), flush=True)
ctx.close()
