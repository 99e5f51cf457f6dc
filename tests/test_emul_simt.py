"""The kernels' CROSS-LANE logic on the CPU: tests/cpu_emul/libemul64.so runs every wavefront as 64 fibers that meet
at the wave-level primitives (__ballot, mbcnt prefix, readfirstlane) — rtk_simt.hpp.  Checked here: the votes sit in
wave-uniform control flow (no divergence error), per-wave candidate lists with the 64-lane stride, packet hand-out by
ballot prefix, and that a 64-lane wave produces exactly what 64 independent lanes produce."""
import numpy as np
import pytest

from conftest import bits, load_golden
import raytracinginoneweekend_amd as rt
import emul
from emul import Emul
from np_mirror import deinterleave


@pytest.fixture(scope="module")
def e64(stock_scene_golden):
    return Emul(stock_scene_golden, lanes=64)


@pytest.mark.parametrize("batch,count", [("primary", 1000), ("secondary", 1000), ("grazing", 1500), ("inside", 700), ("degenerate", 93)])
def test_intersect_in_64_lane_waves_vs_reference(e64, batch, count):
    z = load_golden("rays_stock.npz")
    rays = z[batch + "_rays"][:count]                      # counts that are not multiples of 64: partial last wave
    hit, idx, t = e64.intersect(rays)
    nan = np.isnan(z[batch + "_t"][:count])
    assert np.array_equal(hit, z[batch + "_hit"][:count]) and np.array_equal(idx, z[batch + "_index"][:count])
    assert np.array_equal(np.isnan(t), nan) and np.array_equal(bits(t)[~nan], bits(z[batch + "_t"][:count])[~nan])
    assert emul.lib(64).emul_simt_error() == 0


def test_render_64_lane_waves_equal_single_lanes(stock_scene_golden, e64):
    """Same image bits, same ray count: the refill / packet hand-out across lanes neither loses nor duplicates a sample."""
    e1 = Emul(stock_scene_golden, lanes=1)
    cam = rt.stock_camera(37 / 23)
    for kw in (dict(), dict(run_len=3), dict(tile_w=16, tile_h=4)):
        a, sa = e1.render(cam, 37, 23, 10, 50, seed=5, threads=2, **kw)
        b, sb = e64.render(cam, 37, 23, 10, 50, seed=5, threads=3, **kw)
        assert np.array_equal(bits(a), bits(b)) and sa["rays"] == sb["rays"] and sa["samples"] == sb["samples"] == 37 * 23 * 10
        assert sb["passes"] * 64 >= sb["rays"]             # wave passes x 64 lanes bound the segments traced
    assert emul.lib(64).emul_simt_error() == 0


def test_render_64_lane_ranks(stock_scene_golden, e64):
    W, H, spp, world = 40, 24, 6, 3
    cam = rt.stock_camera(W / H)
    ref, st = e64.render(cam, W, H, spp, 50, seed=2, threads=2)
    tprp = rt.local_tile_count(W, H, 0, world)
    parts = []
    for r in range(world):
        tiles, _ = e64.render_tiles(cam, rt.make_params(W, H, spp, 50, 2, 0.001, r, world), threads=2)
        pad = np.zeros((tprp, 64, 3), np.float32); pad[: len(tiles)] = tiles
        parts.append(pad)
    assert np.array_equal(bits(deinterleave(np.stack(parts), W, H, world)), bits(ref))
    assert emul.lib(64).emul_simt_error() == 0


def test_wave_statistics_are_sane(e64):
    emul.wave_stats(64)
    img, st = e64.render(rt.stock_camera(1.5), 96, 64, 8, 50, seed=1, threads=4)
    s = emul.wave_stats(64)
    # the 64-lane build streams every scene; the stock scene's list is cut into word-local frames there (16 class words of
    # at most 32 entries and one kind-0 word for the three large spheres at its end), the same number for every pass
    assert s["closest_hit_calls"] == st["passes"] and s["filter_words"] % st["passes"] == 0 and 15 <= s["filter_words"] // st["passes"] <= 18
    assert s["class_words"] == (s["filter_words"] // st["passes"] - 1) * st["passes"] and e64.local_words() >= 15
    assert st["rays"] / (64.0 * st["passes"]) > 0.9                      # per-lane refill keeps the lanes busy
    assert 1.0 < s["flush_iters"] / st["passes"] < 12.0


# ---- the staged large-scene path: libemul64.so forces it for every scene (EMUL_FORCE_STAGED), so every 64-lane
# intersect above already went through it; these cases add the chunk boundaries. ----
@pytest.mark.parametrize("n", [1, 63, 127, 128, 129, 160, 255, 256, 257, 1025])
def test_staged_chunks_and_tails(n):
    """Scenes around the 128-sphere chunk size: last chunk of 1, of a partial word, of exactly one chunk."""
    from test_emul_intersect import _odd_scene, tangent_rays
    from oracle import pyoracle as po
    rng = np.random.default_rng(100 + n)
    s = _odd_scene(rng, n)
    m = 1500
    rays = np.concatenate([np.concatenate([rng.uniform(-12, 12, (m, 3)), rng.normal(size=(m, 3))], 1).astype(np.float32),
                           tangent_rays(rng, s, 700, 0.5, 200, 1e-5)])
    o = po.Oracle(); o.set_scene(s)
    ho, io, to = o.intersect(rays, 0.001, np.inf)
    hit, idx, t = Emul(s, lanes=64).intersect(rays)
    nan = np.isnan(to)
    assert np.array_equal(hit, ho) and np.array_equal(idx, io)
    assert np.array_equal(np.isnan(t), nan) and np.array_equal(bits(t)[~nan], bits(to)[~nan])
    assert emul.lib(64).emul_simt_error() == 0


@pytest.mark.parametrize("k,m", [(50, 640), (135, 192)])
def test_staged_large_scenes(k, m):
    """BASELINE.json config 4 (10k and ~73k spheres) through the staged path in 64-lane waves, unsafe rays included
    (all spheres candidates: list flushes and the 64k re-base happen between staged chunks)."""
    from oracle import pyoracle as po
    s = rt.scaled_scene(k=k)
    rng = np.random.default_rng(k)
    ext = float(k)
    rays = np.zeros((m, 6), np.float32)
    rays[:, 0] = rng.uniform(-ext, ext, m); rays[:, 1] = rng.uniform(0.01, 3, m); rays[:, 2] = rng.uniform(-ext, ext, m)
    rays[:, 3:6] = rng.normal(size=(m, 3)); rays[:, 4] *= 0.15
    rays[5:9, 3:6] = (np.nan, 1, 0)
    o = po.Oracle(); o.set_scene(s)
    ho, io, to = o.intersect(rays, 0.001, np.inf)
    hit, idx, t = Emul(s, lanes=64).intersect(rays)
    nan = np.isnan(to)
    assert np.array_equal(hit, ho) and np.array_equal(idx, io)
    assert np.array_equal(np.isnan(t), nan) and np.array_equal(bits(t)[~nan], bits(to)[~nan])
    assert emul.lib(64).emul_simt_error() == 0
