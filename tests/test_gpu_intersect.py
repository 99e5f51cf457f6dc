"""GPU parity of the intersector through the C ABI: rtk_intersect_batch == HittableList::hit of the
reference.  Bar: hit and index bit-exact; t bit-exact too (north_star allows 1e-5 relative; the exact
stage reproduces the reference's operation sequence, so the test holds it to equality)."""
import numpy as np
import pytest

from conftest import bits, load_golden
import raytracinginoneweekend_amd as rt
from oracle import pyoracle as po
from test_oracle_golden import ties_scene

pytestmark = pytest.mark.gpu


def check(got, want_hit, want_idx, want_t):
    hit, idx, t = got
    assert np.array_equal(hit, want_hit), "hit/miss differs in %d rays" % int((hit != want_hit).sum())
    assert np.array_equal(idx, want_idx), "closest-sphere index differs in %d rays" % int((idx != want_idx).sum())
    # t: bit-equal, except that a NaN root (degenerate rays; accepted as a hit by the reference's comparison
    # polarity) may carry a different sign/payload: x86 SSE and gfx950 generate different default NaNs.
    nan = np.isnan(want_t)
    assert np.array_equal(np.isnan(t), nan), "NaN roots differ"
    assert np.array_equal(bits(t)[~nan], bits(want_t)[~nan]), "t differs in %d rays" % int((bits(t)[~nan] != bits(want_t)[~nan]).sum())


@pytest.mark.parametrize("batch", ["primary", "secondary", "tertiary", "random", "grazing", "inside", "degenerate"])
def test_fixed_batches_vs_reference(gpu_ctx, batch):
    z = load_golden("rays_stock.npz")
    check(gpu_ctx.intersect(z[batch + "_rays"]), z[batch + "_hit"], z[batch + "_index"], z[batch + "_t"])


def test_finite_range_vs_reference(gpu_ctx):
    z = load_golden("rays_stock.npz")
    check(gpu_ctx.intersect(z["primary_rays"], 0.5, 9.0), z["primary_tmin0p5_tmax9_hit"], z["primary_tmin0p5_tmax9_index"], z["primary_tmin0p5_tmax9_t"])


def test_tie_rule_vs_reference():
    z = load_golden("rays_ties.npz")
    ctx = rt.Context(0)
    ctx.set_scene(ties_scene(z))
    check(ctx.intersect(z["rays"]), z["hit"], z["index"], z["t"])
    ctx.close()


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 257, 100003])
def test_empty_and_ragged_batches(gpu_ctx, oracle_stock, n):
    rng = np.random.default_rng(n)
    rays = np.concatenate([rng.uniform(-13, 13, (n, 3)), rng.normal(size=(n, 3))], 1).astype(np.float32)
    rays[:, 1] = np.abs(rays[:, 1]) * 0.2
    got = gpu_ctx.intersect(rays)
    assert got[0].shape == (n,)
    check(got, *oracle_stock.intersect(rays))


def test_one_million_random_rays_vs_oracle(gpu_ctx, oracle_stock):
    rng = np.random.default_rng(5)
    n = 1_000_000
    rays = np.zeros((n, 6), np.float32)
    rays[:, 0:3] = rng.uniform(-14, 14, (n, 3)); rays[:, 1] = rng.uniform(0.001, 3, n)
    rays[:, 3:6] = rng.normal(size=(n, 3)) * rng.uniform(1e-3, 1e3, (n, 1))
    # a third start ON sphere surfaces heading outward along the normal + noise (self-intersection regime of
    # the reference's Lambertian bounce, SURVEY.md §7 hard part 3)
    s = oracle_stock.get_scene()
    k = rng.integers(0, 484, n // 3)
    nrm = rng.normal(size=(n // 3, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    c = np.stack([s["cx"][k], s["cy"][k], s["cz"][k]], 1)
    rays[: n // 3, 0:3] = c + nrm * s["radius"][k][:, None]
    rays[: n // 3, 3:6] = nrm + rng.normal(size=(n // 3, 3)) * rng.choice([1.0, 0.1, 1e-3], (n // 3, 1))
    check(gpu_ctx.intersect(rays), *oracle_stock.intersect(rays))


def _odd_scene(rng, n):
    s = {"cx": rng.uniform(-10, 10, n), "cy": rng.uniform(0, 2, n), "cz": rng.uniform(-10, 10, n), "radius": rng.uniform(0.05, 1.0, n)}
    s = {k: v.astype(np.float32) for k, v in s.items()}
    for k in ("mat_r", "mat_g", "mat_b", "roughness", "metallic", "ior"):
        s[k] = np.ones(n, np.float32)
    s["mat_type"] = np.ones(n, np.int32)
    return s


@pytest.mark.parametrize("n", [1, 31, 32, 33, 1000, 1025, 3000])
def test_scene_sizes_lds_and_global_paths(n):
    """n <= 1024 is LDS-resident, beyond that the filter reads the sphere list from global memory."""
    rng = np.random.default_rng(n)
    s = _odd_scene(rng, n)
    o = po.Oracle(); o.set_scene(s)
    ctx = rt.Context(0); ctx.set_scene(s)
    m = 20000
    rays = np.concatenate([rng.uniform(-12, 12, (m, 3)), rng.normal(size=(m, 3))], 1).astype(np.float32)
    check(ctx.intersect(rays), *o.intersect(rays))
    ctx.close()


def test_unsafe_spheres_and_negative_radius():
    """Spheres the filter cannot bound (NaN / inf / huge / tiny) are always sent to the exact stage;
    negative radii (hollow-glass trick) behave as in the reference."""
    rng = np.random.default_rng(3)
    s = _odd_scene(rng, 40)
    s["radius"][3] = -0.7
    s["radius"][5] = np.inf
    s["cx"][7] = np.nan
    s["radius"][9] = 1e-30
    s["cy"][11] = 3e38
    s["radius"][13] = 0.0
    s["radius"][15] = np.nan
    s["cz"][17] = -np.inf
    s["radius"][19] = 1e25
    s["cx"][21] = 2e9
    o = po.Oracle(); o.set_scene(s)
    ctx = rt.Context(0); ctx.set_scene(s)
    m = 50000
    rays = np.concatenate([rng.uniform(-12, 12, (m, 3)), rng.normal(size=(m, 3))], 1).astype(np.float32)
    rays[::7, 0:3] = (s["cx"][9], s["cy"][9], s["cz"][9])            # from the centre of the tiny sphere
    rays[::11, 0:3] = (s["cx"][13], s["cy"][13], s["cz"][13])        # from the centre of the zero-radius sphere
    check(ctx.intersect(rays), *o.intersect(rays))
    ctx.close()


def test_large_scene_config4_vs_oracle():
    """BASELINE.json config 4: ~10k spheres (grid widened to [-50,50)^2), global-memory filter path."""
    s = rt.scaled_scene(k=50)
    o = po.Oracle(); o.set_scene(s)
    ctx = rt.Context(0); ctx.set_scene(s)
    rng = np.random.default_rng(9)
    m = 30000
    rays = np.zeros((m, 6), np.float32)
    rays[:, 0] = rng.uniform(-50, 50, m); rays[:, 1] = rng.uniform(0.01, 4, m); rays[:, 2] = rng.uniform(-50, 50, m)
    rays[:, 3:6] = rng.normal(size=(m, 3)); rays[:, 4] *= 0.2
    check(ctx.intersect(rays), *o.intersect(rays))
    ctx.close()


def test_word_local_frames_vs_staged_walk_and_oracle(monkeypatch):
    """Scenes beyond LDS whose words are class words run the matrix-pipe filter in word-local frames; with
    RTK_NO_LOCAL_FRAMES set the same scene streams through the vector-ALU filter.  Both must give the reference's hits bit
    for bit — also for origins hundreds of units away, where a word is judged as a whole — and the same image."""
    from test_emul_intersect import tangent_rays
    s = rt.scaled_scene(k=50)
    o = po.Oracle(); o.set_scene(s)
    rng = np.random.default_rng(19)
    m = 40000
    rays = np.zeros((m, 6), np.float32)
    rays[:, 0] = rng.uniform(-50, 50, m); rays[:, 1] = rng.uniform(0.01, 4, m); rays[:, 2] = rng.uniform(-50, 50, m)
    rays[:, 3:6] = rng.normal(size=(m, 3)); rays[:, 4] *= 0.2
    rays[::7, 0:3] *= rng.uniform(5, 60, (len(rays[::7]), 1)).astype(np.float32)          # far origins
    rays[::7, 1] -= 300.0                                                                  # ... below the horizon of the ground sphere
    rays[::97, 3] = np.nan
    rays = np.concatenate([rays, tangent_rays(rng, s, 8000, 0.5, 3000, 1e-5, first=1)])
    want = o.intersect(rays)
    results = {}
    for no_local in (False, True):
        if no_local:
            monkeypatch.setenv("RTK_NO_LOCAL_FRAMES", "1")
        else:
            monkeypatch.delenv("RTK_NO_LOCAL_FRAMES", raising=False)
        ctx = rt.Context(0); ctx.set_scene(s)
        lay = ctx.scene_layout()
        assert not lay["lds_resident"]
        assert (lay["class_words"] == 0) if no_local else (lay["class_words"] >= 300)
        check(ctx.intersect(rays), *want)
        img, st = ctx.render(rt.stock_camera(1.5), 96, 64, 8, 50, seed=4)
        results[no_local] = (img, st)
        ctx.close()
    (a, sa), (b, sb) = results[False], results[True]
    assert np.array_equal(bits(a), bits(b)) and sa["rays"] == sb["rays"]
    assert sa["sphere_tests"] < 1.5 * sb["sphere_tests"]          # about as many exact tests as the projection form needs


def test_word_local_frames_100k_spheres_vs_oracle():
    """BASELINE.json configs[3] at its upper end: 99 857 spheres, rows of 316 cut into words of 32 and a rest."""
    s = rt.scaled_scene(k=158)
    assert len(s["cx"]) > 99000
    o = po.Oracle(); o.set_scene(s)
    ctx = rt.Context(0); ctx.set_scene(s)
    assert ctx.scene_layout()["class_words"] >= 3100
    rng = np.random.default_rng(23)
    m = 4000
    rays = np.zeros((m, 6), np.float32)
    rays[:, 0] = rng.uniform(-158, 158, m); rays[:, 1] = rng.uniform(0.01, 4, m); rays[:, 2] = rng.uniform(-158, 158, m)
    rays[:, 3:6] = rng.normal(size=(m, 3)); rays[:, 4] *= 0.1
    check(ctx.intersect(rays), *o.intersect(rays))
    ctx.close()


def test_errors(gpu_ctx):
    ctx = rt.Context(0)
    with pytest.raises(rt.RtkError, match="rtk_set_scene first"):
        ctx.intersect(np.zeros((4, 6), np.float32))
    bad = _odd_scene(np.random.default_rng(0), 4)
    bad = {k: v[:0] for k, v in bad.items()}
    with pytest.raises(rt.RtkError, match="0 spheres"):
        ctx.set_scene(bad)
    ctx.close()


def _tangent_rays(rng, scene, n, dist_lo, dist_hi, rel_eps, first=0):
    """Rays from `dist` away aimed at the silhouette of a random sphere, +- rel_eps radii; unnormalised directions."""
    k = rng.integers(first, len(scene["cx"]), n)
    c = np.stack([scene["cx"][k], scene["cy"][k], scene["cz"][k]], 1).astype(np.float64)
    r = np.abs(scene["radius"][k].astype(np.float64))
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    dist = np.exp(rng.uniform(np.log(dist_lo), np.log(dist_hi), n)) + r
    o = c + u * dist[:, None]
    a = rng.normal(size=(n, 3)); a += (a * u).sum(1, keepdims=True) * -u; a /= np.linalg.norm(a, axis=1, keepdims=True)
    s = np.clip(r * (1 + rel_eps * rng.uniform(-1, 1, n)) / dist, 0, 0.999)
    d = -u * np.sqrt(1 - s * s)[:, None] + a * s[:, None]
    d *= np.exp(rng.uniform(np.log(1e-3), np.log(1e3), (n, 1)))
    return np.concatenate([o, d], 1).astype(np.float32)


@pytest.mark.parametrize("lo,hi", [(0.01, 1), (1, 30), (30, 300), (300, 3000), (3000, 1e5), (1e5, 1e8)])
def test_tangent_rays_at_every_distance(gpu_ctx, oracle_stock, stock_scene_golden, lo, hi):
    """Rays that graze a sphere's silhouette from 0.01 to 1e8 units away: the filter's margin (the matrix pipe's own fp32
    accumulation of cancelling f16 part products, which the CPU build only bounds) must keep every sphere the reference's
    rounding still hits — beyond a few hundred units that is spheres the ray misses by far.  Bit-exact against the oracle."""
    rng = np.random.default_rng(int(lo * 100) + 7)
    rays = np.concatenate([_tangent_rays(rng, stock_scene_golden, 6000, lo, hi, eps, first=1) for eps in (0.0, 1e-6, 1e-3, 0.3)])
    check(gpu_ctx.intersect(rays), *oracle_stock.intersect(rays))


def test_mixed_radii_scene_far_from_the_origin():
    """General words (K = 32, two chained MFMAs) of a scene 3000 units from the origin: the tiles' scale follows the extent of
    the centres, so the margin is at its widest."""
    rng = np.random.default_rng(29)
    s = _odd_scene(rng, 300)
    s["cx"] += np.float32(3000.0); s["cy"] -= np.float32(200.0); s["cz"] += np.float32(1500.0)
    o = po.Oracle(); o.set_scene(s)
    ctx = rt.Context(0); ctx.set_scene(s)
    m = 20000
    org = rng.uniform(-12, 12, (m, 3)) + np.array([3000.0, -200.0, 1500.0])
    rays = np.concatenate([np.concatenate([org, rng.normal(size=(m, 3))], 1).astype(np.float32),
                           _tangent_rays(rng, s, 8000, 0.5, 3000, 1e-5)])
    check(ctx.intersect(rays), *o.intersect(rays))
    ctx.close()
