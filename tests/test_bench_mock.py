"""bench.py's driver logic and JSON contract without a GPU: the context and the renderer are stand-ins that report
fixed counters, torch's CUDA entry points are no-ops.  Checks the one-line JSON (metric, value, roofline, config) for
the headline run, for two frames in flight and for the grid / config-4 options."""
import contextlib
import importlib
import json
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT


class FakeCtx:
    made = 0

    def __init__(self, device):
        FakeCtx.made += 1
        self.frames = 0
        self.accel = 0

    def set_scene(self, scene):
        self.n = len(scene["cx"])

    def set_accel(self, mode):
        self.accel = mode
        return mode

    def scene_layout(self):
        return {"spheres": self.n, "class_words": 15 if self.n == 484 else 0, "lds_resident": self.n <= 1024, "lds_bytes_per_workgroup": 20000}

    def reset_stats(self):
        self.frames = 0

    def get_stats(self):
        f = self.frames
        return {"rays": 1000 * f, "samples": 340 * f, "sphere_tests": 484000 * f, "exact_tests": 2000 * f, "lane_slots": 1020 * f,
                "kernel_ms": 2.0 * f, "total_ms": 0.0, "frames": f, "frames_timed": f}

    def set_stream(self, h):
        pass

    def close(self):
        pass


class FakeRenderer:
    def __init__(self, ctx, W, H, rank, world, device, extra_contexts=()):
        self.ctxs = [ctx, *extra_contexts]
        self.n = 0

    def render(self, cam, spp, depth, seed=0):
        self.ctxs[self.n % len(self.ctxs)].frames += 1
        self.n += 1


@pytest.fixture()
def bench(monkeypatch):
    sys.path.insert(0, ROOT)
    b = importlib.import_module("bench")
    real_device = torch.device
    monkeypatch.setattr(torch, "device", lambda *a, **k: real_device("cpu"))
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(b.rt, "Context", FakeCtx)
    monkeypatch.setattr(b, "ShardedRenderer", FakeRenderer)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    FakeCtx.made = 0
    return b


def run(b, monkeypatch, capsys, *argv):
    monkeypatch.setattr(sys, "argv", ["bench.py", "--no-cpu-baseline", *argv])
    b.main()
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_headline_line(bench, monkeypatch, capsys):
    j = run(bench, monkeypatch, capsys, "--steps", "4", "--warmup", "1")
    assert j["metric"] == "Mrays/s (whole node), final scene 1200x800 500spp" and j["unit"] == "Mrays/s"
    assert j["n_gpus"] == 1 and j["steps"] == 4 and j["warmup"] == 1 and j["higher_is_better"] and j["scaling"] == "strong"
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and j["vs_baseline"] is None and "workload" in j["config"]
    r = j["roofline"]
    assert r["bound"].startswith("simd_issue") and r["unit"] == "TFLOP/s" and r["peak"] == bench.ISSUE_BOUND_PEAK_TFLOPS
    assert abs(r["peak"] - 668.5) < 0.1 and r["fp32_vector_peak"] == bench.FP32_VECTOR_PEAK_TFLOPS          # the VALU side of the mapping: the pipes co-execute
    assert abs(r["frac_of_fp32_vector_peak"] - r["achieved"] / 157.3) < 1e-12
    # 1000 rays per 2 ms launch x 484 spheres x 17 flop
    assert r["rays_per_launch"] == 1000 and abs(r["kernel_ms"] - 2.0) < 1e-9
    assert abs(r["achieved"] - 1000 * 484 * 17 / 2e-3 / 1e12) < 1e-12 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-15
    assert abs(j["config"]["rays_per_sample"] - 1000 / 340) < 1e-12 and j["value"] > 0 and FakeCtx.made == 1


def test_two_frames_in_flight_sum_both_contexts(bench, monkeypatch, capsys):
    j = run(bench, monkeypatch, capsys, "--steps", "5", "--warmup", "2", "--frames-in-flight", "2")
    assert FakeCtx.made == 2 and j["config"]["frames_in_flight"] == 2
    assert j["roofline"]["rays_per_launch"] == 1000           # 5 frames over two contexts, still 1000 rays per launch
    assert abs(j["config"]["rays_per_sample"] - 1000 / 340) < 1e-12


def test_grid_and_config4_are_labelled_as_not_the_headline(bench, monkeypatch, capsys):
    j = run(bench, monkeypatch, capsys, "--steps", "2", "--warmup", "0", "--scene-k", "3", "--accel", "grid", "--width", "64", "--height", "40", "--spp", "4")
    assert "NOT the headline workload" in j["metric"] and "uniform grid" in j["config"]["workload"]
    assert j["roofline"]["frac"] is None and j["roofline"]["kernel"] == "render_kernel<false, true>"
    assert "cpu_baseline" not in j


@pytest.mark.parametrize("argv,cfg,headline", [
    ((), "configs[1]", True),
    (("--width", "3840", "--height", "2160", "--spp", "1000"), "configs[2]", False),
    (("--width", "400", "--height", "225", "--spp", "10"), "configs[0]", False),
    (("--spp", "60"), "no BASELINE config", False),
    (("--depth", "100"), "no BASELINE config", False),
    (("--glass", "--depth", "100"), "configs[4]", False),
    (("--scene-k", "3"), "configs[3]", False),
    (("--mixed", "200"), "no BASELINE config (user scene)", False),
])
def test_every_size_and_scene_names_its_own_config(bench, monkeypatch, capsys, argv, cfg, headline):
    """Only configs[1] exactly (stock scene, list walk, 1200x800, 500 spp, depth 50) carries the headline metric string;
    everything else says which BASELINE config it is (round 2 labelled a 3840x2160x1000 run as the headline)."""
    j = run(bench, monkeypatch, capsys, "--steps", "2", "--warmup", "0", *argv)
    assert (j["metric"] == "Mrays/s (whole node), final scene 1200x800 500spp") == headline
    assert cfg in j["config"]["workload"] and (headline or cfg in j["metric"])
    assert j["roofline"].get("traffic") is None or headline          # counters are replayed for the headline workload only
    w = [a for a in argv if a.isdigit()]
    assert all(x in j["config"]["workload"] for x in w if len(x) >= 3)
