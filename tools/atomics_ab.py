"""tools/atomics_ab.py — what do the u64 accumulator atomics cost?  Kernel time of rank 0 of `world` ranks of the
1200x800x500 frame with the shipped library and with a build whose atomics are removed (-DRTK_EXP_NO_ATOMICS: wrong image,
timing only; the switch was a temporary edit of rtk_render.hpp's flush, `atomicAdd(p, v)` -> nothing, and is not in the tree:
profiles/r03_atomics_ab.txt has the result); run once per library (RTK_LIB).   usage: RTK_LIB=... python tools/atomics_ab.py [world ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracinginoneweekend_amd as rt
W, H, SPP, DEPTH = 1200, 800, 500, 50
ctx = rt.Context(0)
ctx.set_scene(rt.stock_scene())
cam = rt.stock_camera(W / H)
for world in [int(a) for a in sys.argv[1:]] or [1, 8]:
    nlt = rt.lib().rtk_local_tile_count(W, H, 8, 8, 0, world)
    out = torch.zeros((nlt, 64, 3), dtype=torch.float32, device="cuda:0")
    ms = []
    for rep in range(5):
        st = ctx.render_tiles_device(cam, rt.make_params(W, H, SPP, DEPTH, rep, 0.001, 0, world, 8, 8, 0, 0), out.data_ptr(), want_stats=True)
        ms.append(st["kernel_ms"])
    print(os.path.basename(os.environ.get("RTK_LIB", "librtk.so")), "world", world, "kernel ms", ["%.3f" % m for m in ms], "min %.3f" % min(ms[1:]), flush=True)
ctx.close()
