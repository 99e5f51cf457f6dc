#!/bin/bash
# tools/hazard_hunt/run.sh <tag> <frames> <variant> ... — tools/determinism_check.py once per hand-edited build
# (tools/asm_variant.py); a build that renders differing frames is a RESULT, not a failure, so the loop goes on; a
# time-out, abort or fault ends the call (nothing more is started on the GPU).
set -u
TAG=${1:?tag}; FRAMES=${2:?frames}; shift 2
EXTRA=${HUNT_ARGS:-}
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_hazard_hunt.txt
: > $OUT
for V in "$@"; do
  LIB=$PWD/raytracinginoneweekend_amd/variants/librtk_$V.so
  [ -f "$LIB" ] || { echo "missing $LIB" | tee -a $OUT; exit 2; }
  echo "== $V" >> $OUT
  RTK_LIB=$LIB timeout -k 10 120 python tools/determinism_check.py $FRAMES $EXTRA >> $OUT 2>&1
  rc=$?
  echo "$V rc=$rc: $(tail -1 $OUT)"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping after rc=$rc"; exit 1; fi
done
