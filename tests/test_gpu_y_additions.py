"""GPU tests written after this repository's last GPU session (round 1 lost GPU access early): unusual t-ranges,
lists beyond 64k spheres, context reuse, progressive passes, the bench's stream path, the 8-bit output path and the
drop-in executable.  Collected after the tests that have already been green on an MI355X so that, under `pytest -x`,
a surprise here cannot hide them; the uniform-grid tests (test_gpu_z_grid.py) come after these."""
import numpy as np
import pytest

from conftest import bits, load_golden
import raytracinginoneweekend_amd as rt
from oracle import pyoracle as po
from np_mirror import deinterleave
from test_gpu_intersect import check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("t_min,t_max", [(0.0, np.inf), (-5.0, 3.0), (2.0, 1.0), (0.001, np.nan), (np.nan, np.inf), (-np.inf, np.inf)])
def test_unusual_ranges(gpu_ctx, oracle_stock, t_min, t_max):
    z = load_golden("rays_stock.npz")
    rays = np.concatenate([z["primary_rays"][:1500], z["inside_rays"][:800], z["degenerate_rays"]])
    check(gpu_ctx.intersect(rays, t_min, t_max), *oracle_stock.intersect(rays, t_min, t_max))


def test_scene_beyond_64k_spheres():
    """~73k spheres (towards BASELINE.json config 4's 100k): 16-bit candidate offsets are re-based every 64k spheres."""
    s = rt.scaled_scene(k=135)
    o = po.Oracle(); o.set_scene(s)
    ctx = rt.Context(0); ctx.set_scene(s)
    rng = np.random.default_rng(1)
    m = 256
    rays = np.zeros((m, 6), np.float32)
    rays[:, 0] = rng.uniform(-130, 130, m); rays[:, 1] = rng.uniform(0.01, 3, m); rays[:, 2] = rng.uniform(-130, 130, m)
    rays[:, 3:6] = rng.normal(size=(m, 3)); rays[:, 4] *= 0.1
    rays[:8, 3:6] = (np.nan, 1, 0)
    got = ctx.intersect(rays)
    check(got, *o.intersect(rays))
    assert got[1].max() > 65536
    ctx.close()


def test_context_reuse_and_material_none(stock_scene_golden):
    """One context, several scenes one after another (HittableList::clear()/add() on the host side means a new
    rtk_set_scene), and Material::Type::None absorbing every path it meets."""
    ctx = rt.Context(0)
    cam = rt.stock_camera(1.5)
    small = {k: v[:40].copy() for k, v in stock_scene_golden.items()}
    ctx.set_scene(small)
    a, _ = ctx.render(cam, 48, 32, 8, 50, seed=2)
    ctx.set_scene(stock_scene_golden)
    b, sb = ctx.render(cam, 48, 32, 8, 50, seed=2)
    ctx.set_scene(small)
    a2, _ = ctx.render(cam, 48, 32, 8, 50, seed=2)
    assert np.array_equal(bits(a), bits(a2)) and not np.array_equal(a, b)
    none = {k: v.copy() for k, v in stock_scene_golden.items()}
    none["mat_type"][:] = 0
    ctx.set_scene(none)
    img, st = ctx.render(cam, 48, 32, 8, 50, seed=2)
    assert st["rays"] == st["samples"] and not img[-1].any() and img[0].min() > 0
    ctx.close()


def test_progressive_batches_cover_the_same_samples(gpu_ctx):
    """rtk_params::sample_offset (progressive rendering): two passes trace exactly the samples of one frame."""
    cam = rt.stock_camera(1.5)
    full, sf = gpu_ctx.render(cam, 60, 40, 24, 50, seed=9, run_len=4)
    a, sa = gpu_ctx.render(cam, 60, 40, 12, 50, seed=9, run_len=4, sample_offset=0)
    b, sb = gpu_ctx.render(cam, 60, 40, 12, 50, seed=9, run_len=4, sample_offset=12)
    assert sa["rays"] + sb["rays"] == sf["rays"] and sa["samples"] + sb["samples"] == sf["samples"]
    assert np.allclose(a + b, full, rtol=2e-6, atol=1e-6) and not np.array_equal(a, b)
    with pytest.raises(rt.RtkError, match="sample_offset"):
        gpu_ctx.render(cam, 16, 16, 4, 1, sample_offset=-1)


def test_sharded_renderer_single_rank_matches_rtk_render(stock_scene_golden):
    """bench.py's path (ShardedRenderer on its own torch stream: tiles -> gather -> de-interleave) at world size 1."""
    import torch
    from raytracinginoneweekend_amd.distributed import ShardedRenderer
    ctx = rt.Context(0)
    ctx.set_scene(stock_scene_golden)
    W, H = 100, 60
    cam = rt.stock_camera(W / H)
    ref, st = ctx.render(cam, W, H, 16, 50, seed=4)
    sr = ShardedRenderer(ctx, W, H, 0, 1, torch.device("cuda", 0))
    for _ in range(2):                                       # twice: the accumulators must come back zeroed
        sr.render(cam, 16, 50, seed=4)
        sr.synchronize()
        assert np.array_equal(bits(sr.image.cpu().numpy()), bits(ref))
    ctx.set_stream(0)
    ctx.close()


def test_render_u8_is_write_color_binary_of_render(gpu_ctx):
    """rtk_render_u8: the device-side gamma/quantise gives exactly the bytes the reference's
    write_color_binary (Color.hpp:28-48, restated by the oracle) produces from rtk_render's sums."""
    cam = rt.stock_camera(1.5)
    sums, st = gpu_ctx.render(cam, 90, 60, 20, 50, seed=12)
    u8, st8 = gpu_ctx.render_u8(cam, 90, 60, 20, 50, seed=12)
    assert st["rays"] == st8["rays"]
    assert np.array_equal(u8, po.quantise(sums, 20))
    with pytest.raises(rt.RtkError, match="null"):
        rt._check(rt.lib().rtk_render_u8(gpu_ctx._h, None, None, None, None), "rtk_render_u8")


def test_sharded_u8_gather_matches_f32_path(stock_scene_golden):
    """ShardedRenderer(output="u8"): tiles quantised before the gather, bytes de-interleaved on the device."""
    import torch
    from raytracinginoneweekend_amd.distributed import ShardedRenderer
    ctx = rt.Context(0)
    ctx.set_scene(stock_scene_golden)
    W, H, spp = 100, 60, 16
    cam = rt.stock_camera(W / H)
    sr = ShardedRenderer(ctx, W, H, 0, 1, torch.device("cuda", 0))
    sr.render(cam, spp, 50, seed=4)
    sr.synchronize()
    want = po.quantise(sr.image.cpu().numpy(), spp)
    sr.render(cam, spp, 50, seed=4, output="u8")
    sr.synchronize()
    assert np.array_equal(sr.image_u8.cpu().numpy(), want)
    # the byte de-interleave for several ranks, against the numpy restatement of the index math
    world = 3
    tprp = rt.local_tile_count(W, H, 0, world)
    g = torch.randint(0, 256, (world, tprp, 64, 3), dtype=torch.uint8, device="cuda:0")
    img = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    ctx.deinterleave_u8_device(g.data_ptr(), tprp, rt.make_params(W, H, 1, 1, 0, 0.001, 0, world), img.data_ptr())
    ctx.synchronize()
    assert np.array_equal(img.cpu().numpy(), deinterleave(g.cpu().numpy(), W, H, world))
    ctx.set_stream(0)
    ctx.close()


def test_drop_in_executable_writes_the_reference_file(gpu_ctx, tmp_path):
    """raytracinginoneweekend_amd/host/rtiow_amd is the reference's main (main.cpp:20-81) on top of librtk: same argv,
    same image_binary.ppm.  Its payload must be exactly what rtk_render_u8 gives for the same scene, camera and seed,
    in a process of its own (no torch, its own HIP runtime)."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(rt.__file__), "host", "rtiow_amd")
    assert os.path.exists(exe), "build it with python -m raytracinginoneweekend_amd.build"
    W, H, spp, depth = 64, 42, 16, 50
    r = subprocess.run([exe, str(W), str(H), str(spp), str(depth)], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Image Generation took" in r.stderr and "Done." in r.stderr          # the reference's own progress / timing lines
    data = (tmp_path / "image_binary.ppm").read_bytes()
    header = b"P6\n%d %d\n255\n" % (W, H)
    assert data.startswith(header) and len(data) == len(header) + W * H * 3
    got = np.frombuffer(data[len(header):], np.uint8).reshape(H, W, 3)
    want, _ = gpu_ctx.render_u8(rt.stock_camera(float(np.float32(W) / np.float32(H))), W, H, spp, depth, seed=0)
    assert np.array_equal(got, want)
    # progressive passes (rtk_params::sample_offset, sums added on the host) reach the same picture
    r = subprocess.run([exe, str(W), str(H), str(spp), str(depth), "0", "0", "4"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    data2 = (tmp_path / "image_binary.ppm").read_bytes()
    got2 = np.frombuffer(data2[len(header):], np.uint8).reshape(H, W, 3)
    assert np.abs(got2.astype(int) - want.astype(int)).max() <= 1               # four fp32 partial sums instead of one: last-bit rounding only


def test_patched_reference_executable_matches_rtk_render_u8(gpu_ctx, tmp_path):
    """INTEGRATION.md A carried out (tools/prove_integration.py): the REFERENCE's own sources with the two accessors added
    and the loops of main.cpp:64-76 replaced by rtk_render, compiled in the build container against librtk.so
    (oracle/_ref/rtiow_ref_patched).  Its random_scene(), Camera constructor, argv parsing, PPM header and
    write_color_binary are the reference's; the file it writes must equal rtk_render_u8 of the same arguments byte for byte
    — which also says that the reference's scene and camera are the ones this repository's host API builds."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(rt.__file__)), "oracle", "_ref", "rtiow_ref_patched")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/rtiow_ref_patched not built (python tools/prove_integration.py, in the build container)")
    for W, H, spp, depth in ((200, 112, 4, 50), (64, 42, 100, 50)):          # two of the reference's golden-hash argument sets
        r = subprocess.run([exe, str(W), str(H), str(spp), str(depth)], cwd=tmp_path, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "Image Generation took" in r.stderr and "Done." in r.stderr
        data = (tmp_path / "image_binary.ppm").read_bytes()
        header = b"P6\n%d %d\n255\n" % (W, H)
        assert data.startswith(header) and len(data) == len(header) + W * H * 3
        got = np.frombuffer(data[len(header):], np.uint8).reshape(H, W, 3)
        want, _ = gpu_ctx.render_u8(rt.stock_camera(float(np.float32(W) / np.float32(H))), W, H, spp, depth, seed=0)
        assert np.array_equal(got, want)


def test_stream_switch_after_the_old_stream_was_destroyed(stock_scene_golden):
    """rtk_context_set_stream does not touch the stream it leaves (the caller may have destroyed it): render on a caller's
    stream, synchronise, destroy that stream, switch back to the context's own — no error, and the next frame is right."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    ctx = rt.Context(0)
    ctx.set_scene(stock_scene_golden)
    cam = rt.stock_camera(1.5)
    ref, _ = ctx.render(cam, 96, 64, 8, 50, seed=6)
    stream = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(stream)) == 0
    ctx.set_stream(stream.value)
    a, _ = ctx.render(cam, 96, 64, 8, 50, seed=6)                    # synchronises: stats are read back
    assert hip.hipStreamDestroy(stream) == 0
    ctx.set_stream(0)                                                  # old stream is gone
    b, _ = ctx.render(cam, 96, 64, 8, 50, seed=6)
    ctx.close()
    assert np.array_equal(bits(a), bits(ref)) and np.array_equal(bits(b), bits(ref))


def test_image_sides_beyond_65535_are_refused(gpu_ctx):
    """The render kernel keeps a pixel's column and row in one register (16 bits each): wider or taller images are
    refused with RTK_ERR_TOO_LARGE before anything is allocated or launched."""
    import torch
    cam = rt.stock_camera(1.5)
    dummy = torch.zeros(64 * 3, dtype=torch.float32, device="cuda:0")
    for w, h in ((65536, 2), (2, 70000)):
        p = rt.make_params(w, h, 1, 5, 0, 0.001, 0, 1, 8, 8, 0, 0)
        with pytest.raises(rt.RtkError, match="65535"):
            gpu_ctx.render_tiles_device(cam, p, dummy.data_ptr())
    img, st = gpu_ctx.render(cam, 16, 8, 2, 5)                      # the context is still usable
    assert img.shape == (8, 16, 3) and st["samples"] == 16 * 8 * 2
