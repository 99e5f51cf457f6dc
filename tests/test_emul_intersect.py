"""The librtk DEVICE code (filter -> compaction -> exact) compiled for the CPU with one lane per wave
(tests/cpu_emul), checked against the reference fixtures and the oracle WITHOUT a GPU:

* parity: hit / index / t of Intersector::closest_hit == HittableList::hit, bit for bit;
* conservativeness: every (ray, sphere) pair that gets past the reference's discriminant test
  (Sphere3.hpp:37-38, NaN included) is passed on by the filter — in every regime, including origins so far
  away that the reference's own rounding noise accepts spheres the ray misses geometrically, and with the
  hardware approximations (v_rsq / v_rcp / v_sqrt) perturbed by up to +-3 ulp.
"""
import numpy as np
import pytest

from conftest import bits, load_golden
import raytracinginoneweekend_amd as rt
import emul
from emul import Emul
from oracle import pyoracle as po
from test_oracle_golden import ties_scene


def check(scene, rays, t_min=0.001, t_max=np.inf, noises=(0, 3), want=None):
    e = Emul(scene)
    o = po.Oracle()
    o.set_scene(scene)
    ho, io, to = want if want is not None else o.intersect(rays, t_min, t_max)
    dp = o.disc_pass(rays)
    nan = np.isnan(to)
    over = 0.0
    for noise in noises:
        hit, idx, t, flags = e.intersect(rays, t_min, t_max, want_flags=True, ulp_noise=noise)
        assert int(((dp == 1) & (flags == 0)).sum()) == 0, "the filter rejected a sphere the reference lets through"
        assert np.array_equal(hit, ho) and np.array_equal(idx, io)
        assert np.array_equal(np.isnan(t), nan) and np.array_equal(bits(t)[~nan], bits(to)[~nan])
        over = flags.sum() / max(1, dp.sum())
    return over


@pytest.mark.parametrize("batch", ["primary", "secondary", "tertiary", "random", "grazing", "inside", "degenerate"])
def test_fixed_batches_vs_reference(stock_scene_golden, batch):
    z = load_golden("rays_stock.npz")
    over = check(stock_scene_golden, z[batch + "_rays"], want=(z[batch + "_hit"], z[batch + "_index"], z[batch + "_t"]))
    assert over < 1.35          # the filter's extra candidates stay a small fraction of the reference's own


def test_finite_range_and_ties(stock_scene_golden):
    z = load_golden("rays_stock.npz")
    check(stock_scene_golden, z["primary_rays"], 0.5, 9.0, want=(z["primary_tmin0p5_tmax9_hit"], z["primary_tmin0p5_tmax9_index"], z["primary_tmin0p5_tmax9_t"]))
    t = load_golden("rays_ties.npz")
    check(ties_scene(t), t["rays"], want=(t["hit"], t["index"], t["t"]))


def tangent_rays(rng, scene, n, dist_lo, dist_hi, rel_eps, first=0):
    """Rays from `dist` away aimed at the silhouette of a random sphere, +- rel_eps radii; unnormalised."""
    N = len(scene["cx"])
    k = rng.integers(first, N, n)
    c = np.stack([scene["cx"][k], scene["cy"][k], scene["cz"][k]], 1).astype(np.float64)
    r = np.abs(scene["radius"][k].astype(np.float64))
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    dist = np.exp(rng.uniform(np.log(dist_lo), np.log(dist_hi), n)) + r
    o = c + u * dist[:, None]
    w = -u
    a = rng.normal(size=(n, 3)); a -= (a * w).sum(1, keepdims=True) * w; a /= np.linalg.norm(a, axis=1, keepdims=True)
    s = np.clip(r * (1 + rel_eps * rng.uniform(-1, 1, n)) / dist, 0, 0.999)
    d = w * np.sqrt(1 - s * s)[:, None] + a * s[:, None]
    d *= np.exp(rng.uniform(np.log(1e-3), np.log(1e3), (n, 1)))
    return np.concatenate([o, d], 1).astype(np.float32)


@pytest.mark.parametrize("lo,hi", [(0.01, 1), (1, 30), (30, 300), (300, 3000), (3000, 1e5), (1e5, 1e8)])
def test_tangent_rays_at_every_distance(stock_scene_golden, lo, hi):
    """Beyond a few hundred units the reference's c = |oc|^2 - r^2 is pure rounding noise and it 'hits' spheres the
    ray misses by far; the filter must pass all of those too (its margin scales with the same |oc|^2)."""
    rng = np.random.default_rng(int(lo * 100) + 1)
    for eps in (0.0, 1e-6, 1e-3, 0.3):
        check(stock_scene_golden, tangent_rays(rng, stock_scene_golden, 1500, lo, hi, eps, first=1))


def test_random_rays_bulk(stock_scene_golden):
    rng = np.random.default_rng(5)
    n = 150_000
    rays = np.zeros((n, 6), np.float32)
    rays[:, 0:3] = rng.uniform(-14, 14, (n, 3)); rays[:, 1] = rng.uniform(0.001, 3, n)
    rays[:, 3:6] = rng.normal(size=(n, 3)) * rng.uniform(1e-3, 1e3, (n, 1))
    s = stock_scene_golden
    m = n // 3
    k = rng.integers(0, 484, m)
    nrm = rng.normal(size=(m, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    rays[:m, 0:3] = np.stack([s["cx"][k], s["cy"][k], s["cz"][k]], 1) + nrm * s["radius"][k][:, None]   # on surfaces
    rays[:m, 3:6] = nrm + rng.normal(size=(m, 3)) * rng.choice([1.0, 0.1, 1e-3], (m, 1))
    check(stock_scene_golden, rays, noises=(3,))


def _odd_scene(rng, n, scale=1.0, offset=(0, 0, 0)):
    s = {"cx": rng.uniform(-10, 10, n) * scale + offset[0], "cy": rng.uniform(0, 2, n) * scale + offset[1],
         "cz": rng.uniform(-10, 10, n) * scale + offset[2], "radius": rng.uniform(0.05, 1.0, n) * scale}
    s = {k: v.astype(np.float32) for k, v in s.items()}
    for k in ("mat_r", "mat_g", "mat_b", "roughness", "metallic", "ior"):
        s[k] = np.ones(n, np.float32)
    s["mat_type"] = np.ones(n, np.int32)
    return s


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 64, 100, 1025])
def test_scene_sizes_and_tail_word(n):
    rng = np.random.default_rng(n)
    s = _odd_scene(rng, n)
    m = 4000
    rays = np.concatenate([rng.uniform(-12, 12, (m, 3)), rng.normal(size=(m, 3))], 1).astype(np.float32)
    check(s, rays)
    check(s, tangent_rays(rng, s, 2000, 0.5, 200, 1e-5))


@pytest.mark.parametrize("scale,offset", [(1e-3, (0, 0, 0)), (1e3, (0, 0, 0)), (1.0, (1e4, -3e3, 500)), (1e-2, (50, 50, 50)), (1e6, (0, 0, 0)), (1e-6, (0, 0, 0))])
def test_scaled_and_translated_scenes(scale, offset):
    """The margins are relative (to |o| + scene extent), so tiny, huge and far-from-origin scenes stay exact."""
    rng = np.random.default_rng(11)
    s = _odd_scene(rng, 200, scale, offset)
    m = 4000
    rays = np.zeros((m, 6), np.float32)
    rays[:, 0:3] = rng.uniform(-12, 12, (m, 3)) * scale + np.array(offset)
    rays[:, 3:6] = rng.normal(size=(m, 3)) * scale
    check(s, rays)
    check(s, tangent_rays(rng, s, 3000, 0.5 * scale, 500 * scale, 1e-5))


def test_unsafe_spheres_specks_and_negative_radius():
    rng = np.random.default_rng(3)
    s = _odd_scene(rng, 60)
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
    s["radius"][23] = 1e-4         # a speck among unit-size spheres
    s["radius"][25] = 3e-5
    s["radius"][27] = 500.0        # a second giant
    e = Emul(s)
    info, filterable = e.scene_info()
    assert not filterable[[5, 7, 9, 11, 13, 15, 17, 19, 21]].any()
    assert info["r_min"] >= 0.04 and info["c_max"] < 100        # specks and giants were peeled off the bounds
    m = 20000
    rays = np.concatenate([rng.uniform(-12, 12, (m, 3)), rng.normal(size=(m, 3))], 1).astype(np.float32)
    rays[::7, 0:3] = (s["cx"][9], s["cy"][9], s["cz"][9])
    rays[::11, 0:3] = (s["cx"][13], s["cy"][13], s["cz"][13])
    rays[::13, 0:3] = (s["cx"][23], s["cy"][23], s["cz"][23]) + rng.normal(size=(len(rays[::13]), 3)) * 1e-4
    check(s, rays)
    ok = np.array([i for i in range(60) if np.isfinite(s["radius"][i]) and abs(s["radius"][i]) > 0 and np.isfinite(s["cx"][i] + s["cy"][i] + s["cz"][i]) and abs(s["cx"][i]) < 1e6 and abs(s["cy"][i]) < 1e6 and abs(s["radius"][i]) < 1e6])
    sub = {k: v[ok] for k, v in s.items()}
    check(sub, tangent_rays(rng, sub, 5000, 0.01, 100, 1e-5))


def test_many_tiny_spheres_do_not_break_exactness():
    """More specks than the outlier peeling removes: r_min is small, the filter degrades to 'many candidates'
    but results stay exact."""
    rng = np.random.default_rng(4)
    s = _odd_scene(rng, 120)
    s["radius"][:40] = rng.uniform(1e-4, 1e-3, 40).astype(np.float32)
    m = 5000
    rays = np.concatenate([rng.uniform(-12, 12, (m, 3)), rng.normal(size=(m, 3))], 1).astype(np.float32)
    check(s, rays)
    check(s, tangent_rays(rng, s, 5000, 0.01, 50, 1e-4))


def test_large_scene_config4():
    s = rt.scaled_scene(k=50)
    rng = np.random.default_rng(9)
    m = 1500
    rays = np.zeros((m, 6), np.float32)
    rays[:, 0] = rng.uniform(-50, 50, m); rays[:, 1] = rng.uniform(0.01, 4, m); rays[:, 2] = rng.uniform(-50, 50, m)
    rays[:, 3:6] = rng.normal(size=(m, 3)); rays[:, 4] *= 0.2
    check(s, rays, noises=(3,))


# ---- scenes beyond LDS: the matrix-pipe filter in word-local frames (rtk_device.hpp local_term) ----------
def test_local_frames_cut_and_candidate_count():
    """scaled_scene(k=30): 3601 spheres in rows of 60 -> the list is cut at every row end (a word must not wrap to the next
    row: its frame would be wider than kLocalReach radii), all but the three large spheres sit in class words, and the
    filter passes about as many candidates as the vector-ALU projection form it replaces."""
    s = rt.scaled_scene(k=30)
    assert len(s["cx"]) > 1024
    e = Emul(s)
    assert e.local_words() >= 120                      # 60 rows x (32 + 28)
    rng = np.random.default_rng(2)
    m = 2000
    rays = np.zeros((m, 6), np.float32)
    rays[:, 0] = rng.uniform(-30, 30, m); rays[:, 1] = rng.uniform(0.01, 3, m); rays[:, 2] = rng.uniform(-30, 30, m)
    rays[:, 3:6] = rng.normal(size=(m, 3)); rays[:, 4] *= 0.2
    over = check(s, rays)
    emul.lib().emul_local_frames(0)
    try:
        assert Emul(s).local_words() == 0
        over_staged = check(s, rays)
    finally:
        emul.lib().emul_local_frames(1)
    assert over < 1.25 * over_staged + 0.1, (over, over_staged)


@pytest.mark.parametrize("lo,hi", [(0.01, 1), (1, 30), (30, 300), (300, 3000), (3000, 1e5), (1e5, 1e8)])
def test_local_frames_tangent_rays_at_every_distance(lo, hi):
    """Grazing rays from every distance, hardware noise on: near origins are judged by the dot product, origins beyond
    |v|^2 = 2^14 in a word's own units (a bounce off the far side of the ground sphere) by the word as a whole."""
    s = rt.scaled_scene(k=20)
    rng = np.random.default_rng(int(lo * 100) + 7)
    for eps in (0.0, 1e-6, 1e-3, 0.3):
        check(s, tangent_rays(rng, s, 800, lo, hi, eps, first=1))


def test_local_frames_far_origins_do_not_take_every_sphere():
    """A ray that starts ~1000 units away (the far side of the ground sphere) must not turn the short words at the row ends
    into candidates wholesale: that was 400 exact tests for each such ray of the 10 001-sphere scene."""
    s = rt.scaled_scene(k=20)
    rng = np.random.default_rng(8)
    m = 400
    rays = np.zeros((m, 6), np.float32)
    rays[:, 0] = rng.uniform(-900, 900, m); rays[:, 1] = rng.uniform(-1500, -200, m); rays[:, 2] = rng.uniform(-900, 900, m)
    rays[:, 3:6] = rng.normal(size=(m, 3))
    e = Emul(s)
    hit, idx, t, flags = e.intersect(rays, want_flags=True, ulp_noise=3)
    o = po.Oracle(); o.set_scene(s)
    dp = o.disc_pass(rays)
    assert int(((dp == 1) & (flags == 0)).sum()) == 0
    assert flags.sum() < 3 * dp.sum() + 4 * m, (flags.sum(), dp.sum())


@pytest.mark.parametrize("scale,offset", [(1.0, (0, 0, 0)), (1e-3, (0, 0, 0)), (1e3, (0, 0, 0)), (1.0, (1e4, -3e3, 500)), (1e-2, (50, 50, 50)), (3e4, (0, 0, 0)), (1e-5, (0, 0, 0))])
def test_local_frames_scaled_and_translated_class_scenes(scale, offset):
    """A class scene of 1500 random centres (so words are as wide as the scene: frames of sigma_w ~ 1/16 per unit of scale),
    tiny, huge and far from the origin: sigma_w, the folded constants (2^75 sigma_w^2 ...) and the far-origin rule must hold
    at every magnitude the cut admits (sigma_w in 2^-20 .. 2^20); beyond that a scene falls back to the staged walk."""
    rng = np.random.default_rng(21)
    s = _class_scene(rng, 1500, [(0.2, 0.2)], offset=offset, scale=scale)
    e = Emul(s)
    assert e.local_words() >= 40 or scale < 1e-4 or scale > 1e4
    m = 2500
    rays = np.zeros((m, 6), np.float32)
    rays[:, 0:3] = rng.uniform(-12, 12, (m, 3)) * scale + np.array(offset)
    rays[:, 3:6] = rng.normal(size=(m, 3)) * scale
    rays[::9, 0:3] = (rng.uniform(-3000, 3000, (len(rays[::9]), 3)) * scale + np.array(offset)).astype(np.float32)      # far origins
    check(s, rays)
    check(s, tangent_rays(rng, s, 2000, 0.5 * scale, 5000 * scale, 1e-5))


def test_local_frames_words_with_always_candidates_and_ragged_end():
    """Non-filterable entries inside class runs (a giant, a NaN centre, a zero radius) are zero rows with their always bit;
    the list ends in a ragged word."""
    rng = np.random.default_rng(5)
    s = _class_scene(rng, 1237, [(0.2, 0.2)])
    order = np.argsort(s["cx"], kind="stable")                     # neighbours in the list are neighbours in space: narrow words
    for k in s:
        s[k] = s[k][order]
    s["radius"][100] = 900.0; s["cy"][100] = -900.0                # a ground-like giant: peeled off the bounds, always a candidate
    s["cx"][333] = np.nan
    s["radius"][777] = 0.0
    e = Emul(s)
    assert e.local_words() >= 38
    m = 3000
    rays = np.concatenate([np.concatenate([rng.uniform(-12, 12, (m, 3)), rng.normal(size=(m, 3))], 1).astype(np.float32),
                           tangent_rays(rng, {k: np.delete(v, [100, 333, 777]) for k, v in s.items()}, 1500, 0.5, 300, 1e-5)])
    check(s, rays)


def test_local_frames_mixed_scene_keeps_the_staged_walk():
    """Mixed radii: almost no word is a class word, the scene streams through the vector-ALU filter as before."""
    rng = np.random.default_rng(12)
    s = _odd_scene(rng, 1500)
    assert Emul(s).local_words() == 0


def test_candidate_lists_overflow_path(stock_scene_golden):
    """Degenerate (unsafe) rays make all 484 spheres candidates: the 16-entry lane list is flushed 31 times and
    the running closest / NaN behaviour must survive the early exact passes."""
    z = load_golden("rays_stock.npz")
    e = Emul(stock_scene_golden)
    hit, idx, t, flags = e.intersect(z["degenerate_rays"], want_flags=True)
    assert (flags.sum(1) == 484).sum() > 20
    assert e.last_exact >= flags.sum()


def test_scene_beyond_64k_spheres_rebases_the_list():
    """BASELINE.json config 4 goes to 100k spheres: candidate lists hold 16-bit offsets that are re-based every 64k."""
    s = rt.scaled_scene(k=135)                         # ~72.9k spheres
    assert len(s["cx"]) > 70000
    rng = np.random.default_rng(1)
    m = 300
    rays = np.zeros((m, 6), np.float32)
    rays[:, 0] = rng.uniform(-130, 130, m); rays[:, 1] = rng.uniform(0.01, 3, m); rays[:, 2] = rng.uniform(-130, 130, m)
    rays[:, 3:6] = rng.normal(size=(m, 3)); rays[:, 4] *= 0.1
    rays[:20, 3:6] = (np.nan, 1, 0)                    # unsafe rays: all 72.9k spheres become candidates
    check(s, rays, noises=(0,))
    e = Emul(s)
    hit, idx, t = e.intersect(rays)
    assert idx.max() > 65536                           # winners beyond the first 64k block occur


@pytest.mark.parametrize("t_min,t_max", [(0.0, np.inf), (-5.0, 3.0), (2.0, 1.0), (0.001, np.nan), (np.nan, np.inf), (-np.inf, np.inf), (1e-30, 1e30)])
def test_unusual_ranges(stock_scene_golden, t_min, t_max):
    """Sphere3::hit's range tests are plain comparisons (false for NaN): negative, empty, infinite and NaN ranges
    must behave as in the reference."""
    z = load_golden("rays_stock.npz")
    rays = np.concatenate([z["primary_rays"][:1500], z["inside_rays"][:800], z["degenerate_rays"]])
    check(stock_scene_golden, rays, t_min, t_max, noises=(0,))


# ---- class words (SceneDev::wdesc): words whose filterable spheres share the scene's most common (cy, radius) ----------
def _class_scene(rng, n, y_groups, offset=(0.0, 0.0, 0.0), scale=1.0):
    """`n` spheres in list order: consecutive groups take their (cy, radius) from `y_groups` (None = random per sphere)."""
    cx = rng.uniform(-11, 11, n); cz = rng.uniform(-11, 11, n)
    cy = rng.uniform(0.1, 1.5, n); rad = rng.uniform(0.1, 0.5, n)
    per = max(1, n // len(y_groups))
    for g, yr in enumerate(y_groups):
        if yr is not None:
            sl = slice(g * per, n if g == len(y_groups) - 1 else (g + 1) * per)
            cy[sl] = yr[0]; rad[sl] = yr[1]
    s = {"cx": cx * scale + offset[0], "cy": cy * scale + offset[1], "cz": cz * scale + offset[2], "radius": rad * scale}
    s = {k: v.astype(np.float32) for k, v in s.items()}
    for k, v in (("mat_r", 0.5), ("mat_g", 0.5), ("mat_b", 0.5), ("roughness", 1.0), ("metallic", 0.0), ("ior", 1.5)):
        s[k] = np.full(n, v, np.float32)
    s["mat_type"] = np.ones(n, np.int32)
    return s


def test_stock_scene_is_fifteen_class_words_and_a_short_tail(stock_scene_golden):
    """random_scene(): ground (an always-candidate: a zero row in its word), 480 small spheres with r = 0.2, three big ones ->
    words 0..14 are class words, the last 5 spheres a short tail."""
    e = Emul(stock_scene_golden)
    assert e.class_words() == 15


@pytest.mark.parametrize("groups,n,want_words", [([(0.2, 0.2)], 96, 3), ([(0.2, 0.2), None, (1.0, 0.3)], 288, 3), ([None], 64, 0),
                                                 ([(0.2, 0.2)], 40, 1), ([(-0.0, 0.25), (0.0, 0.25)], 64, 2),
                                                 ([(0.2, 0.2), (1.0, 0.3), (1.0, 0.3)], 192, 4)])
def test_class_words_parity_and_conservativeness(groups, n, want_words):
    rng = np.random.default_rng(n + len(groups))
    s = _class_scene(rng, n, groups)
    assert Emul(s).class_words() == want_words
    m = 3000
    rays = np.concatenate([np.concatenate([rng.uniform(-12, 12, (m, 3)), rng.normal(size=(m, 3))], 1).astype(np.float32),
                           tangent_rays(rng, s, 1500, 0.5, 300, 1e-5)])
    check(s, rays)


def test_a_class_word_shares_height_and_radius():
    """A class word's tile holds the monomials of x and z only; cy and r^2 are folded into the ray's side, so both must
    agree bit for bit (-0.0 is another height than 0.0)."""
    rng = np.random.default_rng(3)
    s = _class_scene(rng, 64, [(0.0, 0.25)])
    assert Emul(s).class_words() == 2
    s["cy"][5] = np.float32(-0.0)
    assert Emul(s).class_words() == 1
    s["radius"][40] = np.float32(0.26)
    assert Emul(s).class_words() == 0
    rays = np.concatenate([rng.uniform(-12, 12, (2000, 3)), rng.normal(size=(2000, 3))], 1).astype(np.float32)
    check(s, rays)


@pytest.mark.parametrize("tilt", [0.0, 1e-12, 1e-7, 1e-6, 3e-6, 1e-4, 1e-2])
def test_rays_along_the_y_axis(stock_scene_golden, tilt):
    """The filter's basis needs a horizontal direction: rays within 2^-20 of the y axis test every sphere instead, rays
    just outside that cone use a basis built from a tiny h = |(dx, dz)| — both must stay exact and conservative."""
    rng = np.random.default_rng(int(tilt * 1e13) + 1)
    n = 1500
    o = np.stack([rng.uniform(-11, 11, n), rng.uniform(0.5, 6.0, n), rng.uniform(-11, 11, n)], 1)
    phi = rng.uniform(0, 2 * np.pi, n)
    d = np.stack([tilt * np.cos(phi), -np.ones(n), tilt * np.sin(phi)], 1) * rng.uniform(0.3, 3.0, (n, 1))
    d[::2] *= -1.0                                           # half of them point up (hit nothing or a sphere above)
    # aim a third of them at the silhouette of a small sphere from straight above
    k = rng.integers(1, 481, n // 3)
    c = np.stack([stock_scene_golden["cx"][k], stock_scene_golden["cy"][k], stock_scene_golden["cz"][k]], 1)
    psi = rng.uniform(0, 2 * np.pi, n // 3)
    o[: n // 3] = c + np.stack([0.2 * (1 + rng.uniform(-1e-5, 1e-5, n // 3)) * np.cos(psi), np.full(n // 3, 3.0), 0.2 * np.sin(psi)], 1)
    d[: n // 3, 1] = -np.abs(d[: n // 3, 1])
    rays = np.concatenate([o, d], 1).astype(np.float32)
    check(stock_scene_golden, rays)


def test_translated_class_scene_far_from_the_origin():
    rng = np.random.default_rng(11)
    s = _class_scene(rng, 128, [(0.2, 0.2)], offset=(3000.0, -200.0, 1500.0))
    m = 2000
    o = rng.uniform(-12, 12, (m, 3)) + np.array([3000.0, -200.0, 1500.0])
    rays = np.concatenate([np.concatenate([o, rng.normal(size=(m, 3))], 1).astype(np.float32), tangent_rays(rng, s, 1500, 0.5, 3000, 1e-5)])
    check(s, rays)


def test_tile_layout_helpers():
    """The word tiles of the matrix-pipe filter: the row -> sphere map is a permutation that hands each lane half 16
    consecutive spheres; the f16 conversions of the host build are IEEE binary16 (numpy's); the term map covers every
    monomial with the three part products (1,1), (1,2), (2,1)."""
    import ctypes as C
    lib = emul.lib(1)
    rows = [lib.emul_tile_row_sphere(r) for r in range(32)]
    assert sorted(rows) == list(range(32))
    for rho, sphere in enumerate(rows):
        hf = (rho >> 2) & 1
        assert sphere // 16 == hf                                   # rows 4..7, 12..15, ... belong to lanes 32..63
    # register reg of a lane in half hf is row (reg & 3) + 8 (reg >> 2) + 4 hf: even registers spheres 16 hf + m, odd ones 16 hf + 8 + m
    for hf in range(2):
        for reg in range(16):
            assert rows[(reg & 3) + 8 * (reg >> 2) + 4 * hf] == 16 * hf + 8 * (reg & 1) + (reg >> 1)
    lib.emul_f16_rne.restype = C.c_uint16
    lib.emul_f16_rne.argtypes = [C.c_float]
    lib.emul_f16_value.restype = C.c_float
    lib.emul_f16_value.argtypes = [C.c_uint16]
    rng = np.random.default_rng(11)
    vals = np.concatenate([rng.normal(size=4000) * 10.0 ** rng.uniform(-9, 5, 4000),
                           [0.0, -0.0, 1.0, -1000.0, 0.2, 65504.0, 65519.9, 65520.0, 1e6, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0001, 2.0 ** -14, 6.1e-5,
                            2047.0, 2049.0, 2051.0, np.inf, -np.inf]]).astype(np.float32)
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16)
    for v, h in zip(vals, want):
        bits = lib.emul_f16_rne(float(v))
        assert bits == int(h.view(np.uint16)), (v, bits, h)
        assert np.float32(lib.emul_f16_value(bits)).view(np.uint32) == np.float32(h).view(np.uint32)
    out = (C.c_int * 3)()
    for n_mono, kk in ((5, 16), (9, 32)):
        seen = {}
        for k in range(kk):
            lib.emul_quad_term(k, n_mono, out)
            if out[0] >= 0:
                seen.setdefault(out[0], []).append((out[1], out[2]))
        for m in range(n_mono):
            assert seen[m] == [(0, 0), (0, 1), (1, 0)]
        assert (seen.get(n_mono) == [(0, 0), (1, 0)]) == (n_mono == 9)


def test_every_filter_test_is_exactly_zero_or_one(stock_scene_golden):
    """The clamped value X = -2^48 (D + K) of a candidate must be exactly 1 (the bits are accumulated in fp32): D and K are
    multiples of 2^-48, so X cannot fall inside (0, 1) (rtk_intersect.hpp)."""
    e = Emul(stock_scene_golden)
    z = load_golden("rays_stock.npz")
    emul.wave_stats(1)
    for batch in ("primary", "secondary", "grazing", "inside", "degenerate"):
        e.intersect(z[batch + "_rays"][:4000])
    st = emul.wave_stats(1)
    assert st["filter_words"] > 0 and st["clamp_fractions"] == 0
    info, _ = e.scene_info()
    assert info["sigma"] == 2.0 ** -4 and info["kinds"] == 1        # centres within 16 units; class words only
    assert info["cls_cy"] == np.float32(0.2) / 16 and abs(info["cls_r2"] / (0.04 / 256) - 1) < 1e-4


@pytest.mark.parametrize("n", [1, 5, 8, 9, 31, 32, 33, 40, 41, 64, 72, 73, 485])
def test_last_word_as_short_tail_or_as_tile(n):
    """A last word of 1..8 spheres is tested on the vector ALUs (no tile), a longer one gets a tile with pads."""
    rng = np.random.default_rng(500 + n)
    s = _odd_scene(rng, n)
    m = 1500
    rays = np.concatenate([np.concatenate([rng.uniform(-12, 12, (m, 3)), rng.normal(size=(m, 3))], 1).astype(np.float32),
                           tangent_rays(rng, s, 600, 0.5, 200, 1e-5)])
    emul.wave_stats(1)
    check(s, rays)
    st = emul.wave_stats(1)
    assert (st["tail_tests"] > 0) == (0 < n % 32 <= 8)
