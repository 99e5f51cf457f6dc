"""Kernel time of rank 0 of `world` ranks of the 1200x800x500 frame for the library in RTK_LIB (default schedule).
usage: [RTK_LIB=...] python tools/world_sweep.py [world ...]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracinginoneweekend_amd as rt
W, H, SPP, DEPTH = 1200, 800, 500, 50
ctx = rt.Context(0); ctx.set_scene(rt.stock_scene()); cam = rt.stock_camera(W / H)
for world in [int(a) for a in sys.argv[1:]] or [1, 8]:
    nlt = rt.lib().rtk_local_tile_count(W, H, 8, 8, 0, world)
    out = torch.zeros((nlt, 64, 3), dtype=torch.float32, device="cuda:0")
    ms = []
    for rep in range(4):
        st = ctx.render_tiles_device(cam, rt.make_params(W, H, SPP, DEPTH, rep, 0.001, 0, world, 8, 8, 0, 0), out.data_ptr(), want_stats=True)
        ms.append(st["kernel_ms"])
    print("%s world %d: kernel ms %s, lane utilisation %.4f, drain %.3f" % (os.path.basename(rt.LIB_PATH), world, ["%.2f" % m for m in ms], st["rays"] / st["lane_slots"], st.get("drain_frac", 0.0)), flush=True)
ctx.close()
