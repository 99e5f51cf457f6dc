"""Soak of the uniform-grid closest hit against the oracle (CPU build of the device code; no GPU).
Rays are aimed to travel far through large scenes from origins near the edge of the ball the grid's margins are
sized for — where the reference's own rounding noise (which the grid must reproduce) is largest.
usage: python tools/soak_grid.py [rays per case]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracinginoneweekend_amd as rt
from emul import Emul
from oracle import pyoracle as po

m = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
total = bad = 0
for k in (11, 30, 80, 135):
    s = rt.stock_scene() if k == 11 else rt.scaled_scene(k=k)
    e = Emul(s); g = e.grid_info()
    o = po.Oracle(); o.set_scene(s)
    rng = np.random.default_rng(k)
    D = 2.9 * k
    for frac in (0.05, 0.4, 0.7, 0.745):           # the admissible ball has radius 0.75 D
        org = rng.normal(size=(m, 3)); org[:, 1] = np.abs(org[:, 1]) * 0.05
        org /= np.linalg.norm(org, axis=1, keepdims=True)
        org *= frac * D; org[:, 1] += 0.3
        tgt = np.stack([rng.uniform(-k, k, m), rng.uniform(0.0, 0.45, m), rng.uniform(-k, k, m)], 1)
        d = (tgt - org) * np.exp(rng.uniform(-3, 3, (m, 1)))
        rays = np.concatenate([org, d], 1).astype(np.float32)
        t0 = time.time()
        ho, io, to = o.intersect(rays, 0.001, np.inf)
        hit, idx, t = e.intersect(rays, accel=1)
        n_bad = int((idx != io).sum() + (t.view(np.uint32) != to.view(np.uint32)).sum())
        total += m; bad += n_bad
        print("k=%d N=%d dims=%s origin at %.3f D: %d rays, %d mismatches, %.1f exact tests per ray, hit rate %.2f (%.1fs)" %
              (k, len(s["cx"]), g["dims"], frac, m, n_bad, e.last_exact / m, ho.mean(), time.time() - t0), flush=True)
print("TOTAL", total, "rays,", bad, "mismatches")
sys.exit(1 if bad else 0)
