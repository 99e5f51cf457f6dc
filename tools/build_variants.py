"""Build an experiment variant of librtk.so into raytracinginoneweekend_amd/variants/ (git-ignored, travels with gpurun)
for tools/ab_bench.py: the shipped source with extra compiler flags.  The tree holds ONE code path; a variant exists
only while an experiment is being measured, and its switch is removed once the A/B is settled.

    python tools/build_variants.py ship name=-DRTK_SOMETHING=1,-DOTHER=2 ...
    gpurun -- 'bash tools/gpu_session.sh variants <tag> <spp> ship name'     # parity first, then the A/B
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracinginoneweekend_amd", "csrc")
OUT = os.path.join(ROOT, "raytracinginoneweekend_amd", "variants")
BASE = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include")]
os.makedirs(OUT, exist_ok=True)
for spec in sys.argv[1:] or ["ship"]:
    name, _, flags = spec.partition("=")
    out = os.path.join(OUT, f"librtk_{name}.so")
    subprocess.run(BASE + [f for f in flags.split(",") if f] + [os.path.join(CSRC, "rtk.hip"), "-o", out], check=True)
    print("built", out)
