"""rtk_multi_* (include/rtk.h): one process driving several GPUs — the C/C++ drop-in's way to shard the frame (SURVEY.md
§8b, §8e).  A test box has ONE GPU, so the ranks are several entries of device 0: everything runs except the RCCL calls
(ranks that share a device are gathered with device copies ordered by events; distinct devices take ncclCommInitAll +
one group of ncclSend / ncclRecv — that branch is unmeasured on hardware until the driver's multi-GPU run).  The checks:
the assembled frame is bit-identical to the single-context render for every rank count, the 8-bit output is
write_color_binary of it, per-rank counters add up, and the drop-in executable takes a device list."""
import os
import subprocess

import numpy as np
import pytest

import raytracinginoneweekend_amd as rt
from conftest import bits
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H", [(96, 64), (37, 23)])
def test_multi_ranks_on_one_gpu_equal_the_single_context_render(gpu_ctx, W, H):
    cam = rt.stock_camera(W / H)
    ref, st = gpu_ctx.render(cam, W, H, 24, 50, seed=3)
    for devices in ([0], [0, 0], [0, 0, 0], [0] * 8):
        m = rt.Multi(devices)
        assert not m.uses_rccl                       # one physical device: copy gather
        m.set_scene(rt.stock_scene())
        img, u8, per = m.render(cam, W, H, 24, 50, seed=3, want_u8=True)
        assert np.array_equal(bits(img), bits(ref)), devices
        assert np.array_equal(u8, po.quantise(ref, 24))
        assert sum(d["rays"] for d in per) == st["rays"] and sum(d["samples"] for d in per) == W * H * 24
        assert all(d["frames"] == 1 and d["kernel_ms"] > 0 for d in per)
        # a second frame through the same object (buffers reused), other seed and size
        img2, _, _ = m.render(rt.stock_camera(1.5), 48, 32, 8, 50, seed=9)
        ref2, _ = gpu_ctx.render(rt.stock_camera(1.5), 48, 32, 8, 50, seed=9)
        assert np.array_equal(bits(img2), bits(ref2))
        m.close()


def test_multi_argument_errors():
    with pytest.raises(rt.RtkError):
        rt.Multi([])
    with pytest.raises(rt.RtkError, match="out of range"):
        rt.Multi([0, 7])                             # a one-GPU box has no device 7
    m = rt.Multi([0, 0])
    with pytest.raises(rt.RtkError, match="rtk_set_scene first"):
        m.render(rt.stock_camera(1.5), 16, 16, 1, 1)
    m.close()


def test_drop_in_executable_takes_a_device_list(gpu_ctx, tmp_path):
    """rtiow_amd W H spp depth seed "0,0,0": three ranks from one process; the file equals the one-device file."""
    exe = os.path.join(os.path.dirname(rt.__file__), "host", "rtiow_amd")
    W, H, spp, depth = 64, 42, 16, 50
    r = subprocess.run([exe, str(W), str(H), str(spp), str(depth), "0", "0,0,0"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    data = (tmp_path / "image_binary.ppm").read_bytes()
    header = b"P6\n%d %d\n255\n" % (W, H)
    got = np.frombuffer(data[len(header):], np.uint8).reshape(H, W, 3)
    want, _ = gpu_ctx.render_u8(rt.stock_camera(float(np.float32(W) / np.float32(H))), W, H, spp, depth, seed=0)
    assert np.array_equal(got, want)


def test_drop_in_executable_resumes_from_its_checkpoint(tmp_path):
    """a progressive render interrupted after 2 of 5 passes and resumed by a NEW process writes the same
    file as the uninterrupted run (same passes, same fp32 additions in the same order)."""
    exe = os.path.join(os.path.dirname(rt.__file__), "host", "rtiow_amd")
    args = ["64", "42", "20", "50", "7", "0", "5", "0"]
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(); b.mkdir()
    r = subprocess.run([exe, *args], cwd=a, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    whole = (a / "image_binary.ppm").read_bytes()
    r = subprocess.run([exe, *args, "frame.ckpt", "2"], cwd=b, capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and "stopped after pass 2 of 5" in r.stderr and (b / "frame.ckpt").exists()
    r = subprocess.run([exe, *args, "frame.ckpt", "1"], cwd=b, capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and "stopped after pass 3 of 5" in r.stderr
    r = subprocess.run([exe, *args, "frame.ckpt"], cwd=b, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert (b / "image_binary.ppm").read_bytes() == whole
    # a checkpoint of other parameters is ignored: the frame starts over and still ends in the same file
    r = subprocess.run([exe, "64", "42", "20", "50", "8", "0", "5", "0", "frame.ckpt"], cwd=b, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and (b / "image_binary.ppm").read_bytes() != whole


def test_checkpoint_of_another_scene_setup_is_not_continued_and_partial_runs_leave_no_file(tmp_path):
    """The checkpoint carries a hash of what the pixels depend on (flattened scene, camera, t_min, accelerator): a file
    written for the list walk is not continued by a run that asks for the grid — it starts over at pass 1 instead of
    mixing two renders.  A run that stops early (exit status 3) does not create or truncate image_binary.ppm, and
    progressive options together with a device list are refused instead of being silently ignored."""
    exe = os.path.join(os.path.dirname(rt.__file__), "host", "rtiow_amd")
    args = ["64", "42", "20", "50", "7", "0", "5"]
    r = subprocess.run([exe, *args, "0", "frame.ckpt", "2"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and "stopped after pass 2 of 5" in r.stderr
    assert not (tmp_path / "image_binary.ppm").exists()
    r = subprocess.run([exe, *args, "1", "frame.ckpt", "1"], cwd=tmp_path, capture_output=True, text=True, timeout=300)      # accel 1: other content hash
    assert r.returncode == 3 and "stopped after pass 1 of 5" in r.stderr
    assert not (tmp_path / "image_binary.ppm").exists()
    r = subprocess.run([exe, "64", "42", "20", "50", "7", "0,0", "5"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "single-device" in r.stderr
