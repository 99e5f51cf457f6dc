"""tools/local_frames_soak.py [rounds] — GPU soak of the word-local matrix-pipe walk: class scenes beyond LDS of
varying size, seed, scale and position, rays of every kind (random, from surfaces, tangent at all distances, far origins,
non-finite), hit / index / t against the oracle bit for bit.  Test infrastructure (imports oracle/)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import raytracinginoneweekend_amd as rt
from oracle import pyoracle as po
from test_emul_intersect import tangent_rays

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(2026)
total = 0
t0 = time.time()
for it in range(rounds):
    k = int(rng.integers(17, 70))
    s = rt.scaled_scene(k=k, seed=int(rng.integers(1, 1 << 30)))
    scale = float(rng.choice([1.0, 1.0, 1e-2, 37.0, 1e3]))
    off = rng.choice([0.0, 0.0, 250.0, -4000.0], 3) * scale
    for key in ("cx", "cy", "cz", "radius"):
        s[key] = (s[key].astype(np.float64) * scale).astype(np.float32)
    s["cx"] += np.float32(off[0]); s["cy"] += np.float32(off[1]); s["cz"] += np.float32(off[2])
    n = len(s["cx"])
    m = 12000
    rays = np.zeros((m, 6), np.float32)
    rays[:, 0] = rng.uniform(-k, k, m); rays[:, 1] = rng.uniform(0.01, 4, m); rays[:, 2] = rng.uniform(-k, k, m)
    rays[:, 0:3] = rays[:, 0:3] * scale + off
    rays[:, 3:6] = rng.normal(size=(m, 3)) * scale; rays[:, 4] *= 0.2
    j = rng.integers(1, n, m // 3)                                      # from surfaces
    nrm = rng.normal(size=(len(j), 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    rays[: len(j), 0:3] = np.stack([s["cx"][j], s["cy"][j], s["cz"][j]], 1) + nrm * s["radius"][j][:, None]
    rays[: len(j), 3:6] = nrm + rng.normal(size=(len(j), 3)) * 0.5
    rays[::11, 0:3] = (rng.uniform(-40 * k, 40 * k, (len(rays[::11]), 3)) * scale + off).astype(np.float32)
    rays[::113, 3] = np.nan
    rays[::127, 0] = np.inf
    rays = np.concatenate([rays, tangent_rays(rng, s, 4000, 0.05 * scale, 2000 * scale, 1e-5, first=1)])
    o = po.Oracle(); o.set_scene(s)
    ho, io, to = o.intersect(rays)
    ctx = rt.Context(0); ctx.set_scene(s)
    lay = ctx.scene_layout()
    hit, idx, t = ctx.intersect(rays)
    nan = np.isnan(to)
    ok = np.array_equal(hit, ho) and np.array_equal(idx, io) and np.array_equal(np.isnan(t), nan) and np.array_equal(t.view(np.uint32)[~nan], to.view(np.uint32)[~nan])
    total += len(rays)
    print("round %2d: k %2d, %6d spheres, scale %g, offset %s, local words %d: %s" % (it, k, n, scale, off.tolist(), lay["class_words"], "ok" if ok else "MISMATCH"), flush=True)
    ctx.close()
    if not ok:
        bad = np.nonzero((hit != ho) | (idx != io))[0]
        print("  first mismatches:", bad[:5], rays[bad[:3]].tolist(), idx[bad[:3]], io[bad[:3]])
        sys.exit(1)
print("local-frames soak: %d rays over %d scenes, all equal to the oracle (%.0f s)" % (total, rounds, time.time() - t0))
