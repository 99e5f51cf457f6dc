"""tools/hazard_hunt/intersect_diff.py <n_rays> [--save f.npz | --compare f.npz] — closest hits of a large batch of rays
through rtk_intersect_batch with the library named by RTK_LIB (4 workgroups per CU, grid-stride: the regime in which the
class-word loop failed), saved or compared with a saved result.  A comparison lists the differing rays by the
(workgroup, wave, lane, iteration) that computed them, which is what tells a wave-wide error from a per-lane one."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracinginoneweekend_amd as rt

n = int(sys.argv[1])
rng = np.random.default_rng(7)
ctx = rt.Context(0)
ctx.set_scene(rt.stock_scene())
sc = rt.stock_scene()
# secondary-ray distribution: origins on sphere surfaces, directions normal + unit vector (Material.cpp:11), and camera-like rays
k = rng.integers(0, len(sc["cx"]), n)
nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
c = np.stack([sc["cx"][k], sc["cy"][k], sc["cz"][k]], 1)
u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
rays = np.concatenate([c + nrm * sc["radius"][k][:, None], nrm + u], 1).astype(np.float32)
cam = n // 4
rays[:cam, 0:3] = np.array([13, 2, 3], np.float32) + rng.normal(size=(cam, 3)).astype(np.float32) * 0.05
rays[:cam, 3:6] = -rays[:cam, 0:3] + rng.uniform(-6, 6, (cam, 3)).astype(np.float32)
out = []
for rep in range(3):
    hit, idx, t = ctx.intersect(rays)
    out.append((hit.copy(), idx.copy(), t.copy()))
if "--save" in sys.argv:
    f = sys.argv[sys.argv.index("--save") + 1]
    for rep in range(1, 3):
        assert np.array_equal(out[0][1], out[rep][1]) and np.array_equal(out[0][2].view(np.uint32), out[rep][2].view(np.uint32)), "reference build is not deterministic"
    np.savez(f, hit=out[0][0], idx=out[0][1], t=out[0][2])
    print("saved", f, "hits", int(out[0][0].sum()), "of", n)
else:
    z = np.load(sys.argv[sys.argv.index("--compare") + 1])
    n_cu = 256
    stride = n_cu * 4 * 256
    for rep in range(3):
        bad = np.nonzero((out[rep][1] != z["idx"]) | (out[rep][2].view(np.uint32) != z["t"].view(np.uint32)))[0]
        print(f"rep {rep}: {len(bad)} of {n} rays differ")
        if len(bad) == 0:
            continue
        th = bad % stride
        wg, wave, lane, it = th // 256, (th % 256) // 64, th % 64, bad // stride
        key = (it.astype(np.int64) * (stride // 64) + th // 64)          # one wave pass
        uniq, counts = np.unique(key, return_counts=True)
        print(f"   in {len(uniq)} wave passes; lanes wrong per affected pass: min {counts.min()} median {int(np.median(counts))} max {counts.max()}")
        print(f"   lane halves: lanes 0-31: {(lane < 32).sum()}, lanes 32-63: {(lane >= 32).sum()};  waves of the workgroup: {np.bincount(wave, minlength=4).tolist()}")
        kinds = {"missed hit (ref hit, got miss)": ((z['idx'][bad] >= 0) & (out[rep][1][bad] < 0)).sum(),
                 "false hit (ref miss, got hit)": ((z['idx'][bad] < 0) & (out[rep][1][bad] >= 0)).sum(),
                 "other sphere": ((z['idx'][bad] >= 0) & (out[rep][1][bad] >= 0)).sum()}
        print("  ", kinds)
        for j in bad[:12]:
            print(f"   ray {j}: wg {(j % stride) // 256} wave {((j % stride) % 256) // 64} lane {j % 64} iter {j // stride}: ref idx {z['idx'][j]} t {z['t'][j]:.6g}  got idx {out[rep][1][j]} t {out[rep][2][j]:.6g}  (ref word {z['idx'][j] // 32 if z['idx'][j] >= 0 else -1}, got word {out[rep][1][j] // 32 if out[rep][1][j] >= 0 else -1})")
