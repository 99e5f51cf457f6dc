"""The packed-f16 projection filter (RTK_FILTER_FORM=2, an experiment build of the library; DESIGN.md §4.1 "form 2")
on the CPU build of the device code: 5 instructions per ray-sphere test instead of 9, at the price of binary16
rounding inside the filter — which must still never reject a sphere the reference lets through, and must leave hit,
index and t bit-exact.  binary16 results are additionally perturbed by +-2 ulp (4x the rounding the margins assume)."""
import numpy as np
import pytest

from conftest import bits, load_golden
import raytracinginoneweekend_amd as rt
from emul import Emul
from oracle import pyoracle as po
from test_emul_intersect import _odd_scene, tangent_rays
from test_oracle_golden import ties_scene


def check(scene, rays, t_min=0.001, t_max=np.inf, noises=(0, 2), want=None):
    e = Emul(scene, lanes="f2")
    o = po.Oracle(); o.set_scene(scene)
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
def test_f16_fixed_batches_vs_reference(stock_scene_golden, batch):
    z = load_golden("rays_stock.npz")
    over = check(stock_scene_golden, z[batch + "_rays"], want=(z[batch + "_hit"], z[batch + "_index"], z[batch + "_t"]))
    assert over < 1.6           # extra candidates stay a fraction of the reference's own (1.35 for the f32 form)


def test_f16_ranges_and_ties(stock_scene_golden):
    z = load_golden("rays_stock.npz")
    check(stock_scene_golden, z["primary_rays"], 0.5, 9.0, want=(z["primary_tmin0p5_tmax9_hit"], z["primary_tmin0p5_tmax9_index"], z["primary_tmin0p5_tmax9_t"]))
    t = load_golden("rays_ties.npz")
    check(ties_scene(t), t["rays"], want=(t["hit"], t["index"], t["t"]))


@pytest.mark.parametrize("lo,hi", [(0.01, 1.0), (1.0, 100.0), (100.0, 1e4), (1e4, 1e8)])
def test_f16_tangent_rays_at_every_distance(stock_scene_golden, lo, hi):
    rng = np.random.default_rng(int(lo * 100) + 1)
    for eps in (1e-6, 1e-4, 1e-2):
        check(stock_scene_golden, tangent_rays(rng, stock_scene_golden, 4000, lo, hi, eps, first=1))


def test_f16_random_rays_bulk(stock_scene_golden):
    rng = np.random.default_rng(42)
    m = 60000
    rays = np.concatenate([rng.uniform(-14, 14, (m, 3)) * np.array([1, 0.2, 1]) + np.array([0, 1.0, 0]), rng.normal(size=(m, 3))], 1).astype(np.float32)
    check(stock_scene_golden, rays, noises=(2,))


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 64, 100, 1025])
def test_f16_scene_sizes(n):
    rng = np.random.default_rng(n)
    s = _odd_scene(rng, n)
    m = 3000
    rays = np.concatenate([rng.uniform(-12, 12, (m, 3)), rng.normal(size=(m, 3))], 1).astype(np.float32)
    check(s, rays)
    check(s, tangent_rays(rng, s, 2000, 0.5, 200, 1e-5))


@pytest.mark.parametrize("scale,offset", [(1e-3, (0, 0, 0)), (1e3, (0, 0, 0)), (1.0, (1e4, -3e3, 500)), (1e-2, (50, 50, 50)), (1e6, (0, 0, 0)), (1e-6, (0, 0, 0))])
def test_f16_scaled_and_translated_scenes(scale, offset):
    """Filter space is recentred and scaled by a power of two, so binary16's 5 exponent bits are not an issue."""
    rng = np.random.default_rng(11)
    s = _odd_scene(rng, 200, scale, offset)
    m = 4000
    rays = np.zeros((m, 6), np.float32)
    rays[:, 0:3] = rng.uniform(-12, 12, (m, 3)) * scale + np.array(offset)
    rays[:, 3:6] = rng.normal(size=(m, 3)) * scale
    check(s, rays)
    check(s, tangent_rays(rng, s, 3000, 0.5 * scale, 500 * scale, 1e-5))


def test_f16_unsafe_spheres_specks_and_far_origins():
    rng = np.random.default_rng(3)
    s = _odd_scene(rng, 60)
    s["radius"][3] = -0.7; s["radius"][5] = np.inf; s["cx"][7] = np.nan; s["radius"][9] = 1e-30; s["cy"][11] = 3e38
    s["radius"][13] = 0.0; s["radius"][15] = np.nan; s["cz"][17] = -np.inf; s["radius"][19] = 1e25; s["cx"][21] = 2e9
    s["radius"][23] = 1e-4; s["radius"][25] = 3e-5; s["radius"][27] = 500.0
    m = 20000
    rays = np.concatenate([rng.uniform(-12, 12, (m, 3)), rng.normal(size=(m, 3))], 1).astype(np.float32)
    rays[::7, 0:3] *= 1e3          # origins beyond binary16's range in filter space: those lanes test everything
    rays[::11, 0:3] *= 1e6
    check(s, rays)


def test_f16_large_scene_and_many_specks():
    s = rt.scaled_scene(k=50)      # 10k spheres: reach / r_min = 350, the filter still culls but with a small g
    rng = np.random.default_rng(9)
    m = 800
    rays = np.zeros((m, 6), np.float32)
    rays[:, 0] = rng.uniform(-50, 50, m); rays[:, 1] = rng.uniform(0.01, 4, m); rays[:, 2] = rng.uniform(-50, 50, m)
    rays[:, 3:6] = rng.normal(size=(m, 3)); rays[:, 4] *= 0.2
    check(s, rays, noises=(2,))
    s = _odd_scene(rng, 120)
    s["radius"][:40] = rng.uniform(1e-4, 1e-3, 40).astype(np.float32)     # r_min too small for binary16: the scene keeps the f32 form
    rays = np.concatenate([rng.uniform(-12, 12, (4000, 3)), rng.normal(size=(4000, 3))], 1).astype(np.float32)
    check(s, rays)


def test_f16_render_is_bit_identical_to_the_f32_filter(stock_scene_golden):
    """The filter only selects candidates; the exact stage decides.  Same image bits, same ray count."""
    cam = rt.stock_camera(1.5)
    a, sa = Emul(stock_scene_golden).render(cam, 60, 40, 6, 50, seed=3, threads=4)
    b, sb = Emul(stock_scene_golden, lanes="f2").render(cam, 60, 40, 6, 50, seed=3, threads=4)
    assert np.array_equal(bits(a), bits(b)) and sa["rays"] == sb["rays"]
    assert sb["exact_tests"] < 1.35 * sa["exact_tests"]


def test_f16_in_64_lane_waves(stock_scene_golden):
    """The pair loop's per-wave pieces (candidate lists with the 64-lane stride, early flushes when a lane's list
    fills, the uniform always-mask) under 64 fibers per wave."""
    import emul
    z = load_golden("rays_stock.npz")
    e = Emul(stock_scene_golden, lanes="f2x64")
    for batch, count in (("secondary", 1000), ("grazing", 700), ("degenerate", 93)):
        hit, idx, t = e.intersect(z[batch + "_rays"][:count])
        nan = np.isnan(z[batch + "_t"][:count])
        assert np.array_equal(hit, z[batch + "_hit"][:count]) and np.array_equal(idx, z[batch + "_index"][:count])
        assert np.array_equal(bits(t)[~nan], bits(z[batch + "_t"][:count])[~nan])
    cam = rt.stock_camera(37 / 23)
    a, sa = Emul(stock_scene_golden).render(cam, 37, 23, 6, 50, seed=5, threads=2)
    b, sb = e.render(cam, 37, 23, 6, 50, seed=5, threads=3)
    assert np.array_equal(bits(a), bits(b)) and sa["rays"] == sb["rays"]
    assert emul.lib("f2x64").emul_simt_error() == 0
