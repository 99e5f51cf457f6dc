"""Build librtk.so (HIP kernels + C ABI) in-tree for gfx950.  `python -m raytracinginoneweekend_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librtk.so")
HOST_BIN = os.path.join(HERE, "host", "rtiow_amd")
HOST_CAPI = os.path.join(HERE, "librtiow_host.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the exact intersection stage and the shading arithmetic must round every fp32
# operation separately, as the reference build does (rtk_device.hpp header comment).
# No -mllvm options: the filter's matrix instructions are asm statements that name ordinary VGPRs as their results
# (rtk_intersect.hpp mfma_pair), so the experimental -amdgpu-mfma-vgpr-form of round 2 is no longer needed.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function"]


def _stale(target: str, sources) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, "rtk.hip")]
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "rtk.h")]   # every header counts
    if force or _stale(LIB, deps):
        cmd = [HIPCC, *HIP_FLAGS, "-shared", "-I", os.path.join(ROOT, "include"), *srcs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return LIB


def build_known_bad(force: bool = False, verbose: bool = False) -> str:
    """variants/librtk_known_bad.so — TEST INFRASTRUCTURE, never loaded by the product: the shipped kernel source with one
    scalar move between the two MFMAs of a filter word (profiles/r03_hazard_notes.md), the form that computed wrong
    candidate words on MI355X.  tests/test_gpu_render.py::test_known_bad_loop_form_is_caught renders with it and expects
    the parity checks to fail; tests/test_cabi.py expects the code-object lint to flag it."""
    out = os.path.join(HERE, "variants", "librtk_known_bad.so")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "rtk.h")]
    if force or _stale(out, deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        cmd = [HIPCC, *HIP_FLAGS, "-DRTK_EXP_SCALAR_BETWEEN_MFMAS=1", "-shared", "-I", os.path.join(ROOT, "include"), os.path.join(CSRC, "rtk.hip"), "-o", out]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return out


def build_host(force: bool = False, verbose: bool = False) -> str:
    """The drop-in C++ main (reference CLI, reference scene API) linked against librtk.so."""
    hdir = os.path.join(HERE, "host")
    src = os.path.join(hdir, "main.cpp")
    if not os.path.exists(src):
        return ""
    deps = [src, os.path.join(ROOT, "include", "rtk.h")] + [os.path.join(ROOT, "include", "rtiow", f) for f in os.listdir(os.path.join(ROOT, "include", "rtiow"))]
    if force or _stale(HOST_BIN, deps + [LIB]):
        cmd = ["g++", "-std=c++20", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-I",
               os.path.join(ROOT, "include", "rtiow"), src, "-o", HOST_BIN, "-L", HERE, "-lrtk", f"-Wl,-rpath,{HERE}",
               "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return HOST_BIN


def build_host_capi(force: bool = False, verbose: bool = False) -> str:
    """C view of the C++ host scene API for Python (tests, bench.py): librtiow_host.so.  CPU only."""
    src = os.path.join(HERE, "host", "host_capi.cpp")
    inc = os.path.join(ROOT, "include", "rtiow")
    deps = [src, os.path.join(ROOT, "include", "rtk.h")] + [os.path.join(inc, f) for f in os.listdir(inc)]
    if force or _stale(HOST_CAPI, deps):
        cmd = ["g++", "-std=c++20", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), "-I", inc,
               src, "-o", HOST_CAPI]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return HOST_CAPI


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_lib(force, verbose)
    build_host_capi(force, verbose)
    build_host(force, verbose)
    build_known_bad(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
