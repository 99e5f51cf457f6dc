"""Build A/B variants of librtk.so into raytracinginoneweekend_amd/variants/ (git-ignored, travels with gpurun) for
tools/ab_bench.py.  Each variant is the shipped source with extra -D flags.  REVIEW a variant's memory accesses
before running it on a GPU: a faulting kernel closes GPU access for the round (it happened in round 1).

    python tools/build_variants.py            # builds the set below
    gpurun -- 'python tools/ab_bench.py 100 2'
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracinginoneweekend_amd", "csrc")
OUT = os.path.join(ROOT, "raytracinginoneweekend_amd", "variants")
BASE = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include")]
VARIANTS = {
    "ship": [],                                    # projection filter, LDS broadcast operands
    "form0": ["-DRTK_FILTER_FORM=0"],              # discriminant filter (11 VALU / test)
    "smem": ["-DRTK_SMEM=1"],                      # filter operands by scalar loads from the GLOBAL copy (SGPR operands)
    "f16": ["-DRTK_FILTER_FORM=2"],                # packed-f16 projection filter: 3 v_pk_fma_f16 + v_dot2_f32_f16 + v_alignbit per test (5 instead of 9)
    "f16smem": ["-DRTK_FILTER_FORM=2", "-DRTK_SMEM=1"],   # both: pair records by s_load_dwordx16 (8 spheres per load), v_pk_fma_f16 with SGPR sources, no LDS in the loop
    "f16smem_exactg": ["-DRTK_FILTER_FORM=2", "-DRTK_SMEM=1", "-DRTK_EXACT_GLOBAL=1"],   # ... and no scene data in LDS at all (8 KiB of candidate lists per workgroup)
    "exactg": ["-DRTK_EXACT_GLOBAL=1"],            # exact-stage operands from global memory: 16 instead of 24 KiB LDS per workgroup (7 waves/SIMD)
    "exactg7": ["-DRTK_EXACT_GLOBAL=1", "-DRTK_WAVES_PER_EU=7"],   # ... with the register budget of 7 waves per SIMD (71 VGPRs, no spills)
    "staged": ["-DRTK_STAGED_LARGE=1"],            # scenes > 1024 spheres: stream the list through wave-private LDS chunks (only differs for such scenes)
}
os.makedirs(OUT, exist_ok=True)
for name, flags in VARIANTS.items():
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    out = os.path.join(OUT, f"librtk_{name}.so")
    subprocess.run(BASE + flags + [os.path.join(CSRC, "rtk.hip"), "-o", out], check=True)
    print("built", out)
