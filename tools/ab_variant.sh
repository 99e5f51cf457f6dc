#!/bin/bash
# tools/ab_variant.sh <tag> <variant ...> — one bench line per library (the shipped one first) on the 10k-sphere scene (and, with
# SCENES="50 158", the 100k one); variants are raytracinginoneweekend_amd/variants/librtk_<name>.so (tools/build_variants.py)
TAG=$1; shift
mkdir -p gpurun_out
for v in shipped "$@"; do
  lib=$PWD/raytracinginoneweekend_amd/librtk.so; [ $v != shipped ] && lib=$PWD/raytracinginoneweekend_amd/variants/librtk_$v.so
  for k in ${SCENES:-50}; do
    spp=16; [ $k = 158 ] && spp=2; [ $k = 0 ] && spp=100
    RTK_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --scene-k $k --spp $spp > gpurun_out/${TAG}_$v.json 2> gpurun_out/${TAG}_$v.err || { echo FAILED $v $k; tail -5 gpurun_out/${TAG}_$v.err; exit 1; }
    python - $v $k gpurun_out/${TAG}_$v.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1]); r = j["roofline"]
print("%-16s k=%-4s %8.1f Mrays/s  kernel %8.2f ms  exact tests/ray %5.2f  lane util %.3f  words %d" % (sys.argv[1], sys.argv[2], j["value"], r["kernel_ms"], r["exact_tests_per_ray"], r["lane_utilisation"], r["class_words"]), flush=True)
PY
  done
done
