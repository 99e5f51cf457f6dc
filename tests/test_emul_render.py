"""Shading, RNG, camera and the persistent render loop of librtk, run on the CPU through the one-lane-per-wave
emulation build (tests/cpu_emul) — the same source the GPU kernels are compiled from.  Mirrors the GPU tests in
tests/test_gpu_scatter.py / test_gpu_render.py so that the device logic is checked even without a GPU."""
import numpy as np
import pytest

from conftest import bits, load_golden
import raytracinginoneweekend_amd as rt
import emul
from emul import Emul
from oracle import pyoracle as po
from np_mirror import deinterleave, philox4x32_10, unit_vector_to_uniforms
from test_oracle_golden import _scatter_scene, glass_draw_alignment


def test_philox_known_answers():
    kat = [((0, 0, 0, 0), 0, (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, 0xffffffffffffffff, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), 0xa4093822 | (0x299f31d0 << 32), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, seed, want in kat:
        assert tuple(int(x) for x in emul.philox(seed, np.array([ctr], np.uint32))[0]) == want
    ctr = np.random.default_rng(0).integers(0, 2**32, (2000, 4), dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(emul.philox(0x123456789abcdef, ctr), philox4x32_10(ctr, (0x89abcdef, 0x1234567)))


def test_hit_record_vs_reference(stock_scene_golden):
    z = load_golden("sphere_hit.npz")
    s = dict(stock_scene_golden)
    s["mat_type"] = np.ones(484, np.int32)
    s["roughness"] = np.zeros(484, np.float32)
    out9 = z["out9"]
    out = Emul(s).scatter(z["rays"], z["sphere"], out9[:, 6].copy(), np.full((len(out9), 4), 0.25, np.float32))
    assert np.array_equal(bits(out[:, 0:3]), bits(out9[:, 0:3]))
    assert np.array_equal(out[:, 3:6], out9[:, 3:6]) and (out[:, 6] == 1).all()


def test_scatter_branches_vs_reference():
    z = load_golden("scatter.npz")
    # metal
    scene, n = _scatter_scene(z, "metal")
    d = z["metal_draws"].astype(np.float64)
    rad = np.linalg.norm(d, axis=1)
    u1, u2 = unit_vector_to_uniforms((d / rad[:, None]).astype(np.float32))
    u = np.stack([u1, u2, (rad ** 3).astype(np.float32), np.zeros(n, np.float32)], 1)
    out = Emul(scene).scatter(z["metal_rays"], np.arange(n, dtype=np.int32), z["metal_t"], u)
    want = z["metal_out7"]
    mirror = scene["roughness"] == 0
    assert np.array_equal(bits(out[mirror]), bits(want[mirror]))
    assert np.array_equal(bits(out[:, 0:3]), bits(want[:, 0:3])) and np.allclose(out[:, 3:6], want[:, 3:6], atol=3e-5)
    assert (out[:, 6] != want[:, 6]).mean() < 0.002
    # lambertian
    scene, n = _scatter_scene(z, "lambertian")
    u1, u2 = unit_vector_to_uniforms(z["lambertian_draws"])
    out = Emul(scene).scatter(z["lambertian_rays"], np.arange(n, dtype=np.int32), z["lambertian_t"], np.stack([u1, u2, 0 * u1, 0 * u1], 1))
    want = z["lambertian_out7"]
    assert np.array_equal(bits(out[:, 0:3]), bits(want[:, 0:3])) and np.allclose(out[:, 3:6], want[:, 3:6], atol=3e-5)
    # glass
    scene, n = _scatter_scene(z, "glass")
    o = po.Oracle(); o.set_scene(scene)
    sphere = np.arange(n, dtype=np.int32)
    q, consumed = glass_draw_alignment(o, z["glass_rays"], sphere, z["glass_t"], z["glass_draws"])
    u = np.zeros((n, 4), np.float32); u[:, 3] = q[:, 6]
    out = Emul(scene).scatter(z["glass_rays"], sphere, z["glass_t"], u)
    same = np.all(bits(out) == bits(z["glass_out7"]), axis=1)
    assert same.mean() > 0.999 and same[~consumed].all()


def test_primary_rays_match_reference_camera_model():
    W, H = 200, 112
    cam = rt.stock_camera(W / H)
    rng = np.random.default_rng(2)
    xys = np.stack([rng.integers(0, W, 3000), rng.integers(0, H, 3000), rng.integers(0, 500, 3000)], 1).astype(np.int32)
    got = emul.primary_rays(cam, W, H, xys, 99)
    # every ray starts on the lens disk and passes through its pixel's footprint on the focus plane
    lens = got[:, 0:3] - cam[0:3]
    assert (np.linalg.norm(lens, axis=1) <= cam[21] * (1 + 1e-5)).all() and np.abs(lens @ cam[18:21]).max() < 1e-5
    target = got[:, 0:3] + got[:, 3:6]
    rel = target - cam[3:6]
    s = rel @ cam[6:9] / (cam[6:9] @ cam[6:9]); t = rel @ cam[9:12] / (cam[9:12] @ cam[9:12])
    assert (s * (W - 1) >= xys[:, 0] - 1e-3).all() and (s * (W - 1) <= xys[:, 0] + 1 + 1e-3).all()
    y = H - 1 - xys[:, 1]
    assert (t * (H - 1) >= y - 1e-3).all() and (t * (H - 1) <= y + 1 + 1e-3).all()


@pytest.fixture(scope="module")
def stock_emul(stock_scene_golden):
    return Emul(stock_scene_golden)


def test_image_matches_cpu_reference(stock_emul):
    """The emulated kernel's 200-spp image against the 2000-spp golden: two independent reference renders at 250 spp
    are 39.6 dB apart; a biased sampler or a wrong bounce would fall far below."""
    z = load_golden("image_120x80_2000spp.npz")
    W, H, spp = 120, 80, 200
    img, st = stock_emul.render(rt.stock_camera(W / H), W, H, spp, 50, seed=2025)
    psnr = po.psnr_u8(po.quantise(img, spp), po.quantise(z["rgb_sum"], 2000))
    print("emulated kernel, %d spp vs golden: %.2f dB" % (spp, psnr))
    assert psnr >= 39.0
    assert np.allclose((img / spp).mean((0, 1)), (z["rgb_sum"] / 2000).mean((0, 1)), rtol=1e-2)
    gseg = float(z["segments"]) / (W * H * 2000)
    assert st["samples"] == W * H * spp and abs(st["rays"] / st["samples"] - gseg) < 0.01 * gseg


def test_scheduling_independence(stock_emul):
    """Integer accumulation: 1 thread or 8, 1 rank or 3, any tile shape -> the same bits and the same ray count."""
    W, H, spp = 37, 23, 12
    cam = rt.stock_camera(W / H)
    ref, st = stock_emul.render(cam, W, H, spp, 50, seed=3, threads=1)
    for threads in (2, 8):
        img, s2 = stock_emul.render(cam, W, H, spp, 50, seed=3, threads=threads)
        assert np.array_equal(bits(img), bits(ref)) and s2["rays"] == st["rays"]
    for world in (2, 3):
        tprp = rt.local_tile_count(W, H, 0, world)
        parts, rays = [], 0
        for r in range(world):
            tiles, s3 = stock_emul.render_tiles(cam, rt.make_params(W, H, spp, 50, 3, 0.001, r, world), threads=4)
            pad = np.zeros((tprp, 64, 3), np.float32); pad[: len(tiles)] = tiles
            parts.append(pad); rays += s3["rays"]
        assert np.array_equal(bits(deinterleave(np.stack(parts), W, H, world)), bits(ref)) and rays == st["rays"]
    for tw, th in ((16, 4), (64, 1)):
        img, _ = stock_emul.render(cam, W, H, spp, 50, seed=3, threads=4, tile_w=tw, tile_h=th)
        assert np.array_equal(bits(img), bits(ref))


def test_depth_and_run_length(stock_emul):
    cam = rt.stock_camera(1.5)
    img0, st0 = stock_emul.render(cam, 24, 16, 4, 0)
    assert st0["rays"] == 0 and st0["samples"] == 24 * 16 * 4 and not img0.any()
    img1, st1 = stock_emul.render(cam, 24, 16, 4, 1)
    assert st1["rays"] == st1["samples"] and img1[0].min() > 0 and not img1[-1].any()
    a, sa = stock_emul.render(cam, 24, 16, 20, 50, seed=8, run_len=20)
    b, sb = stock_emul.render(cam, 24, 16, 20, 50, seed=8, run_len=1)
    c, sc = stock_emul.render(cam, 24, 16, 20, 50, seed=8, run_len=7)
    # samples are summed as 32.32 fixed point: the partition into runs does not change a bit
    assert sa["rays"] == sb["rays"] == sc["rays"] and np.array_equal(bits(a), bits(b)) and np.array_equal(bits(a), bits(c))


def test_near_zero_threshold_is_exact_in_fp32():
    """rtk_device.hpp replaces the reference's `(double)|x| < 1e-8` (Vector3.cpp:138-141) by `|x| <= 0x322BCC77`."""
    u = np.arange(0x322BC000, 0x322BD000, dtype=np.uint32)
    x = u.view(np.float32)
    assert np.array_equal(x.astype(np.float64) < 1e-8, u <= 0x322BCC77)


def test_material_none_absorbs(stock_scene_golden):
    """Material::Type::None (make_material, Material.cpp:49-52,56-63): scatter returns false -> the path is black."""
    s = {k: v.copy() for k, v in stock_scene_golden.items()}
    s["mat_type"][:] = 0
    e = Emul(s)
    img, st = e.render(rt.stock_camera(1.5), 24, 16, 8, 50, seed=1)
    assert not img[-1].any() and img[0].min() > 0                 # ground rows black, sky rows lit
    assert st["rays"] == st["samples"]                            # every path ends at its first hit or in the sky
    out = e.scatter(np.array([[0, 5, 0, 0, -1, 0]], np.float32), np.array([481], np.int32), np.array([3.0], np.float32), np.zeros((1, 4), np.float32))
    assert out[0, 6] == 0.0


def test_progressive_batches_cover_the_same_samples(stock_emul):
    """rtk_params::sample_offset: two passes over [0,12) and [12,24) trace exactly the samples of one 24-sample frame
    (same RNG counters -> same ray count; sums equal up to the fp32 rounding of the per-pass totals)."""
    cam = rt.stock_camera(1.5)
    full, sf = stock_emul.render(cam, 30, 20, 24, 50, seed=9, run_len=4)
    a, sa = stock_emul.render(cam, 30, 20, 12, 50, seed=9, run_len=4, sample_offset=0)
    b, sb = stock_emul.render(cam, 30, 20, 12, 50, seed=9, run_len=4, sample_offset=12)
    assert sa["rays"] + sb["rays"] == sf["rays"] and sa["samples"] + sb["samples"] == sf["samples"]
    assert np.allclose(a + b, full, rtol=2e-6, atol=1e-6)
    assert not np.array_equal(a, b)
