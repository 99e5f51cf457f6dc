#!/bin/bash
# tools/profile.sh <tag> [bench.py arguments] — rocprofv3 evidence for bench.py's workload on the GPU box.
# One run per pass (rocprofv3 does not split counters over passes; --pmc is never combined with other traces).
# Results land under gpurun_out/prof_<tag>/; tools/summarize_profile.py turns them into profiles/<tag>_*.
set -u
TAG=${1:-r01}
shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# further arguments go to bench.py (e.g. --glass --depth 100 for BASELINE.json configs[4])
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
echo "== kernel trace + stats" 
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
grep '"metric"' $OUT/trace.log > $OUT/trace_bench.json
i=0
for PMC in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE GRBM_COUNT" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES"
do
  i=$((i+1))
  echo "== pmc pass $i: $PMC"
  timeout -s KILL 120 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -o pmc$i -- $CMD > $OUT/pmc$i.log 2>&1
  rc=$?
  if grep -q "error code 38" $OUT/pmc$i.log; then echo "pass $i: too many counters"; fi
  # a failed pass may have left the GPU in a bad state: start nothing more on it in this call
  if [ $rc -ne 0 ]; then echo "pass $i rc=$rc; stopping"; tail -3 $OUT/pmc$i.log; exit 1; fi
done
find $OUT -name "*.csv" | head -40
