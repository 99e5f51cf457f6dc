#!/bin/bash
# tools/hazard_hunt/depth1_parts.sh <variant ...> — the known-bad loop form (-DRTK_EXP_SCALAR_BETWEEN_MFMAS=1) with parts of the depth-1
# path removed (-DRTK_HUNT_*: apply tools/hazard_hunt/depth1_parts.patch to the csrc of the commit that added it, build with
# tools/build_variants.py name=flag,flag): FRAMES frames at path depth 1, distinct (image hash : exact tests) pairs with their counts
mkdir -p gpurun_out
for v in "$@"; do
  echo "== $v"
  RTK_LIB=$PWD/raytracinginoneweekend_amd/variants/librtk_$v.so timeout -k 10 200 python tools/determinism_check.py ${FRAMES:-8} --depth 1 --count-tests 2>&1 | awk "/^[0-9]+ /{print \$2} END{}" | sort | uniq -c | sort -rn | head -4
done
