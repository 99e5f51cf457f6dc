"""tools/hazard_hunt/frame_diff.py --save f.npy | --compare f.npy — one 1200x800x60 frame with the library named by RTK_LIB;
a comparison reports how the differing pixels cluster in the 8x8 tiles (a packet = one tile x one run of samples is
traced by ONE wave: a wave-wide error shows as many pixels of one tile, a per-lane error as isolated pixels)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracinginoneweekend_amd as rt
ctx = rt.Context(0)
ctx.set_scene(rt.stock_scene())
W, H, spp = 1200, 800, 60
cam = rt.stock_camera(W / H)
if "--save" in sys.argv:
    img, st = ctx.render(cam, W, H, spp, 50, seed=11)
    np.save(sys.argv[sys.argv.index("--save") + 1], img)
    print("saved; rays", st["rays"])
else:
    ref = np.load(sys.argv[sys.argv.index("--compare") + 1])
    for rep in range(3):
        img, st = ctx.render(cam, W, H, spp, 50, seed=11)
        d = np.any(img.reshape(H, W, 3) != ref.reshape(H, W, 3), axis=2)
        ys, xs = np.nonzero(d)
        tiles = (ys // 8) * (W // 8) + xs // 8
        ut, cnt = np.unique(tiles, return_counts=True)
        print(f"frame {rep}: rays {st['rays']}; {d.sum()} pixels differ in {len(ut)} tiles; pixels per affected tile: "
              f"1: {(cnt == 1).sum()}, 2-3: {((cnt >= 2) & (cnt <= 3)).sum()}, 4-15: {((cnt >= 4) & (cnt <= 15)).sum()}, 16+: {(cnt >= 16).sum()}, max {cnt.max() if len(cnt) else 0}")
        # magnitude of the differences (in samples' worth of colour: a sample contributes at most ~1 per channel)
        diff = np.abs(img.reshape(H, W, 3) - ref.reshape(H, W, 3)).max(axis=2)[d]
        if len(diff):
            print(f"   |difference| of a pixel's colour sum: median {np.median(diff):.3g}, max {diff.max():.3g}")
        # where in the image
        if len(ys):
            print(f"   rows: {np.percentile(ys, [0, 25, 50, 75, 100]).astype(int).tolist()}  columns: {np.percentile(xs, [0, 25, 50, 75, 100]).astype(int).tolist()}")
