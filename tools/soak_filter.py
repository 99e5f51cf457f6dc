"""Randomised soak of the filter's conservativeness and the intersector's parity on the CPU build of the device
code (tests/cpu_emul): random scenes (sizes, scales, offsets, radius spreads), random + tangent + surface rays,
+-3 ulp perturbation of the hardware approximations.  Prints a summary line per scene; exits non-zero on any miss.
usage: python tools/soak_filter.py [n_scenes] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from emul import Emul
from oracle import pyoracle as po
from test_emul_intersect import tangent_rays

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
build = 1
noise = 3        # f32 ulps added to every hardware approximation
pairs = rays_total = 0
t0 = time.time()
for it in range(n_scenes):
    n = int(rng.choice([1, 3, 17, 33, 100, 484, 700]))
    scale = float(10.0 ** rng.uniform(-4, 4))
    offset = rng.normal(size=3) * scale * float(rng.choice([0, 1, 30, 1000]))
    spread = float(rng.choice([1.0, 3.0, 30.0]))
    s = {"cx": rng.uniform(-10, 10, n) * scale + offset[0], "cy": rng.uniform(-2, 2, n) * scale + offset[1], "cz": rng.uniform(-10, 10, n) * scale + offset[2],
         "radius": np.exp(rng.uniform(np.log(0.1), np.log(0.1 * spread * 3), n)) * scale}
    if rng.random() < 0.3 and n > 4:
        s["radius"][0] = 1000 * scale; s["cy"][0] = offset[1] - 1000 * scale - 2 * scale     # a ground sphere
    if rng.random() < 0.2 and n > 8:
        s["radius"][5] = 1e-4 * scale                                                          # a speck
    if rng.random() < 0.2 and n > 8:
        s["radius"][7] *= -1                                                                   # hollow-glass trick
    s = {k: v.astype(np.float32) for k, v in s.items()}
    for k in ("mat_r", "mat_g", "mat_b", "roughness", "metallic", "ior"):
        s[k] = np.ones(n, np.float32)
    s["mat_type"] = np.ones(n, np.int32)
    m = 3000
    rays = np.zeros((m, 6), np.float32)
    rays[:, 0:3] = rng.uniform(-15, 15, (m, 3)) * scale * float(rng.choice([1, 1, 10, 1000])) + offset
    rays[:, 3:6] = rng.normal(size=(m, 3)) * float(10.0 ** rng.uniform(-6, 6))
    k = rng.integers(0, n, m // 3)
    nrm = rng.normal(size=(m // 3, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    rays[: m // 3, 0:3] = np.stack([s["cx"][k], s["cy"][k], s["cz"][k]], 1) + nrm * np.abs(s["radius"][k])[:, None]
    finite = np.isfinite(s["radius"])
    allr = np.concatenate([rays] + [tangent_rays(rng, s, 1500, lo * scale, hi * scale, eps) for lo, hi, eps in ((0.01, 10, 1e-6), (10, 1e3, 1e-4), (1e3, 1e6, 1e-2))])
    e = Emul(s, lanes=build); o = po.Oracle(); o.set_scene(s)
    ho, io, to = o.intersect(allr)
    dp = o.disc_pass(allr)
    hit, idx, t, flags = e.intersect(allr, want_flags=True, ulp_noise=noise)
    miss = int(((dp == 1) & (flags == 0)).sum())
    nan = np.isnan(to)
    ok = miss == 0 and np.array_equal(hit, ho) and np.array_equal(idx, io) and np.array_equal(np.isnan(t), nan) and np.array_equal(t.view(np.uint32)[~nan], to.view(np.uint32)[~nan])
    pairs += allr.shape[0] * n; rays_total += allr.shape[0]
    if not ok or it % 20 == 0:
        print("scene %d n=%d scale=%.1e  misses=%d parity=%s  cand/refpass=%.2f" % (it, n, scale, miss, ok, flags.sum() / max(1, dp.sum())), flush=True)
    if not ok:
        sys.exit(1)
print("soak ok: %d scenes, %d rays, %d ray-sphere pairs, 0 filter misses, parity exact, %.0fs" % (n_scenes, rays_total, pairs, time.time() - t0))
