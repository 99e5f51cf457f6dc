"""The CPU build of the device code under AddressSanitizer + UBSan (host code only; tests/cpu_emul/asan_driver.cpp):
list walk, staged large-scene path and uniform grid over scenes of 40 ... 67k spheres with finite, degenerate and
far-away rays, four t-ranges, and small three-rank renders.  A kernel index that leaves its buffer on the GPU would
fault the device; here it fails this test instead."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("exe", ["asan_driver", "asan_driver_staged"])
def test_device_code_is_clean_under_asan_and_ubsan(exe, tmp_path):
    subprocess.run(["make", "-C", os.path.join(HERE, "cpu_emul"), "asan"], check=True, capture_output=True)
    r = subprocess.run([os.path.join(HERE, "cpu_emul", exe)], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "all clean" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr
