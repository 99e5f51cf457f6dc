#!/bin/bash
# tools/pmc_compare.sh <tag> <variant> [<variant> ...] — issue/wait/scalar-cache counters of the render kernel for builds under
# raytracinginoneweekend_amd/variants/ (one rocprofv3 --pmc run per pass and build; nothing else is traced).
# Results: gpurun_out/pmc_<tag>/<variant>_<pass>/...csv
set -u
TAG=${1:?tag}; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
for V in "$@"; do
  export RTK_LIB=$PWD/raytracinginoneweekend_amd/variants/librtk_$V.so
  [ -f "$RTK_LIB" ] || { echo "missing $RTK_LIB"; exit 2; }
  i=0
  for PMC in \
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
    "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES SQC_TC_DATA_READ_REQ SQC_TC_STALL SQ_INSTS_VALU SQ_WAVES" \
    "SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY"
  do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $PMC --output-format csv -d $OUT/${V}_$i -o pmc -- $CMD > $OUT/${V}_$i.log 2>&1
    rc=$?
    if [ $rc -ne 0 ]; then echo "$V pass $i rc=$rc; stopping"; tail -3 $OUT/${V}_$i.log; exit 1; fi
    echo "$V pass $i ok"
  done
done
