"""The oracle (oracle/cpu_ref.cpp) against everything the real reference pins for this path.

The reference has no tests of its own; what pins it is determinism.  The fixtures under
tests/golden/ were recorded by tools/make_golden.py from the reference compiled UNMODIFIED from
/root/reference (oracle/Makefile -> oracle/_ref/).  CPU only.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLD, bits, load_golden
from oracle import pyoracle as po

HASHES = json.load(open(os.path.join(GOLD, "ppm_hashes.json")))
# BASELINE.md §2 / SURVEY.md §8c, measured by the survey from the compiled reference
SURVEY_HASHES = {
    "400 225 10 50": "f87897196e761b05544c5dcd940b34128661b94e2ff10402ac4859aae7e31c34",
    "200 112 4 50": "1d58f051c7c2a48d1368952cd2d8ec17b411f645fbd0fcfbe15c9670e87705a3",
    "64": "9168f3ddab25588700201eac913e1cc0b73a57bf043a6ffc9b69e24fdbecc602",
}


def test_recorded_hashes_are_the_surveys():
    assert HASHES == SURVEY_HASHES


@pytest.mark.parametrize("argv", ["64", "200 112 4 50", "400 225 10 50"])
def test_oracle_ppm_hash(argv):
    """Scene + camera + RNG order + intersection + scatter + quantiser, end to end, bit for bit."""
    assert po.reference_main_ppm_sha256(argv.split()) == HASHES[argv]


def test_bench_pins_config1_rays_and_hash():
    """bench.py's cpu_baseline leg prices the reference's run of BASELINE.json configs[0] with a pinned ray count and
    checks its PPM hash: both are what the hash-pinned oracle gives for that argv."""
    import hashlib, importlib.util, os
    spec = importlib.util.spec_from_file_location("bench_consts", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    src = open(spec.origin).read()
    consts = {}
    for name in ("C1_ARGV", "C1_RAYS", "C1_PPM_SHA256"):
        line = next(l for l in src.splitlines() if l.startswith(name + " = "))
        exec(line, consts)
    W, H, spp, depth, aspect = po.cli_dims(consts["C1_ARGV"])
    o = po.Oracle()
    o.random_scene()
    _, u8, seg = o.render_rows(po.stock_camera(aspect), W, H, spp, depth)
    assert seg == consts["C1_RAYS"]
    assert hashlib.sha256(po.ppm_bytes(u8)).hexdigest() == consts["C1_PPM_SHA256"] == HASHES[" ".join(consts["C1_ARGV"])]


def test_oracle_scene_is_reference_scene(stock_scene_golden):
    o = po.Oracle()
    assert o.random_scene() == 484
    s = o.get_scene()
    for k in po.SCENE_FIELDS:
        assert np.array_equal(s[k].view(np.uint32), stock_scene_golden[k].view(np.uint32)), k
    assert list(np.bincount(s["mat_type"])) == [0, 381, 75, 28]


def test_oracle_camera_fields():
    z = load_golden("camera_stock.npz")
    for name in ("a1p5", "a400_225", "a16_9"):
        got = po.stock_camera(float(z[name + "_aspect"]))
        assert np.array_equal(bits(got), bits(z[name])), name


@pytest.mark.parametrize("batch", ["primary", "secondary", "tertiary", "random", "grazing", "inside", "degenerate"])
def test_oracle_list_hit(oracle_stock, batch):
    z = load_golden("rays_stock.npz")
    hit, idx, t = oracle_stock.intersect(z[batch + "_rays"])
    assert np.array_equal(hit, z[batch + "_hit"])
    assert np.array_equal(idx, z[batch + "_index"])
    assert np.array_equal(bits(t), bits(z[batch + "_t"]))


def test_oracle_list_hit_finite_range(oracle_stock):
    z = load_golden("rays_stock.npz")
    hit, idx, t = oracle_stock.intersect(z["primary_rays"], 0.5, 9.0)
    assert np.array_equal(hit, z["primary_tmin0p5_tmax9_hit"])
    assert np.array_equal(idx, z["primary_tmin0p5_tmax9_index"])
    assert np.array_equal(bits(t), bits(z["primary_tmin0p5_tmax9_t"]))


def ties_scene(z):
    n = len(z["cx"])
    s = {k: z[k] for k in ("cx", "cy", "cz", "radius")}
    for k in ("mat_r", "mat_g", "mat_b", "roughness", "metallic", "ior"):
        s[k] = np.ones(n, np.float32)
    s["mat_type"] = np.ones(n, np.int32)
    return s


def test_oracle_tie_rule():
    """Equal roots: the LATER list entry wins (Sphere3.hpp:43 strict `t_max < root`)."""
    z = load_golden("rays_ties.npz")
    o = po.Oracle()
    o.set_scene(ties_scene(z))
    hit, idx, t = o.intersect(z["rays"])
    assert np.array_equal(hit, z["hit"]) and np.array_equal(idx, z["index"]) and np.array_equal(bits(t), bits(z["t"]))
    # the axis rays hit spheres 0 and 1 at the same t: index 1 must win, never 0
    assert set(np.unique(z["index"][:400])) == {1}


def test_oracle_sphere_hit(oracle_stock):
    z = load_golden("sphere_hit.npz")
    out = oracle_stock.sphere_hit(z["rays"], z["sphere"])
    assert np.array_equal(bits(out), bits(z["out9"]))


def _scatter_scene(z, name):
    n = len(z[name + "_radius"])
    m = z[name + "_mat7"]
    rough = np.minimum(m[:, 3], np.float32(1.0))          # the factories clamp (Material.cpp:70,80,91)
    return {"cx": z[name + "_centers"][:, 0].copy(), "cy": z[name + "_centers"][:, 1].copy(), "cz": z[name + "_centers"][:, 2].copy(),
            "radius": z[name + "_radius"], "mat_r": m[:, 0].copy(), "mat_g": m[:, 1].copy(), "mat_b": m[:, 2].copy(),
            "roughness": rough, "metallic": np.where(m[:, 6] == 3, np.float32(0), m[:, 4]).astype(np.float32), "ior": m[:, 5].copy(),
            "mat_type": m[:, 6].astype(np.int32)}, n


def glass_draw_alignment(o, rays, sphere, t, draws):
    """Material.cpp:40 draws from G0 only when refraction is possible: find those rays by probing."""
    n = len(t)
    q0 = np.zeros((n, 7), np.float32)
    q1 = np.zeros((n, 7), np.float32)
    q1[:, 6] = 1.0
    consumed = np.any(bits(o.scatter(rays, sphere, t, q0)) != bits(o.scatter(rays, sphere, t, q1)), axis=1)
    q = np.zeros((n, 7), np.float32)
    q[consumed, 6] = draws[: int(consumed.sum())]
    return q, consumed


@pytest.mark.parametrize("name", ["lambertian", "metal", "glass"])
def test_oracle_scatter(name):
    """Material::scatter per branch, fed with the draws the reference consumed (recorded in a twin process)."""
    z = load_golden("scatter.npz")
    scene, n = _scatter_scene(z, name)
    o = po.Oracle()
    o.set_scene(scene)
    sphere = np.arange(n, dtype=np.int32)
    rays, t, draws = z[name + "_rays"], z[name + "_t"], z[name + "_draws"]
    q = np.zeros((n, 7), np.float32)
    if name == "lambertian":
        q[:, 0:3] = draws
    elif name == "metal":
        q[:, 3:6] = draws
    else:
        q, consumed = glass_draw_alignment(o, rays, sphere, t, draws)
        assert 0 < consumed.sum() < n          # both total internal reflection and refraction occur
    out = o.scatter(rays, sphere, t, q)
    assert np.array_equal(bits(out), bits(z[name + "_out7"]))


def test_oracle_samplers():
    z = load_golden("samplers.npz")
    for kind, key in ((0, "in_sphere"), (1, "unit_vector"), (2, "in_disk")):
        o = po.Oracle()
        assert np.array_equal(bits(o.sample(kind, 4096)), bits(z[key])), key


def test_oracle_quantiser():
    z = load_golden("quantise.npz")
    assert np.array_equal(po.quantise(z["sums"], int(z["spp"])), z["bytes"])


def test_golden_image_is_consistent():
    """The 2000-spp golden is 8 decorrelated 250-spp oracle renders; its first part must be reproducible."""
    z = load_golden("image_120x80_2000spp.npz")
    assert int(z["spp"]) == 2000 and z["rgb_sum"].shape == (80, 120, 3)
    o = po.Oracle()
    o.random_scene()
    o.discard(0, 7)                       # part 0's stream offsets in tools/make_golden.py
    sums, _, _ = o.render_rows(po.stock_camera(120 / 80), 120, 80, int(z["spp_part"]), int(z["depth"]), row_begin=0, row_end=4, want_u8=False)
    assert np.array_equal(bits(sums), bits(z["first_part_sum"][:4]))
    # and a 250-spp part sits at the expected noise distance from the full image (~39.6 dB at 250 vs 250)
    a = po.quantise(z["first_part_sum"], 250)
    b = po.quantise(z["rgb_sum"], 2000)
    assert 38.0 < po.psnr_u8(a, b) < 46.0
