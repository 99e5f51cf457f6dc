"""First GPU contact: parity of the intersector, a small render vs the oracle, and a C2-sized timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracinginoneweekend_amd as rt
from oracle import pyoracle as po

ctx = rt.Context(0)
scene = rt.stock_scene()
ctx.set_scene(scene)
print("scene", ctx.n, flush=True)

# philox KAT (Random123 kat_vectors)
out = ctx.philox(0, np.zeros((1, 4), np.uint32))
print("philox zero:", [hex(x) for x in out[0]])

o = po.Oracle(); o.random_scene()
rng = np.random.default_rng(1)
n = 200000
rays = np.zeros((n, 6), np.float32)
rays[:, 0:3] = rng.uniform(-12, 12, (n, 3)); rays[:, 1] = rng.uniform(0.0, 3.0, n)
rays[:, 3:6] = rng.normal(size=(n, 3))
rays[: n // 2, 0:3] = (13, 2, 3)
t0 = time.time(); h, i, t = ctx.intersect(rays); t1 = time.time()
ho, io, to = o.intersect(rays); t2 = time.time()
print("intersect gpu %.3fs cpu %.3fs" % (t1 - t0, t2 - t1))
print("hit mismatches", int((h != ho).sum()), "idx mismatches", int((i != io).sum()), "t bit mismatches", int((t.view(np.uint32) != to.view(np.uint32)).sum()), "hits", int(ho.sum()), flush=True)

W, H, spp, depth = 120, 80, 100, 50
cam = rt.stock_camera(W / H)
img, st = ctx.render(cam, W, H, spp, depth, seed=7)
print("render small", st, flush=True)
sums, u8, seg = o.render_rows(po.stock_camera(W / H), W, H, spp, depth)
g8 = po.quantise(img, spp)
print("psnr vs cpu (both %d spp): %.2f dB" % (spp, po.psnr_u8(g8, u8)), "cpu segs/sample %.3f gpu %.3f" % (seg / (W * H * spp), st["rays"] / st["samples"]))
print("mean cpu %.4f gpu %.4f" % (sums.mean() / spp, img.mean() / spp), flush=True)
img2, st2 = ctx.render(cam, W, H, spp, depth, seed=7)
print("deterministic:", np.array_equal(img, img2))

W, H, spp = 1200, 800, 50
cam = rt.stock_camera(W / H)
for rep in range(2):
    img, st = ctx.render(cam, W, H, spp, depth, seed=1)
    mr = st["rays"] / (st["kernel_ms"] * 1e3)
    print("C2/10: %s  -> %.1f Mrays/s, roofline %.3f, lane util %.3f" % (st, mr, mr * 1e6 * 17 * ctx.n / 157.3e12, st["rays"] / max(1, st["lane_slots"])), flush=True)
