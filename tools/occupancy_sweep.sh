#!/bin/bash
mkdir -p gpurun_out
for b in 4 3 2 1; do
  RTK_EXP_BLOCKS_PER_CU=$b timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --scene-k 50 --spp 16 > gpurun_out/occ_$b.json 2> gpurun_out/occ_$b.err || { echo FAILED $b; tail -3 gpurun_out/occ_$b.err; exit 1; }
  python - $b gpurun_out/occ_$b.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1]); r = j["roofline"]
print("workgroups per CU %s: %8.1f Mrays/s  kernel %8.2f ms  lane util %.3f" % (sys.argv[1], j["value"], r["kernel_ms"], r["lane_utilisation"]), flush=True)
PY
done
