#!/bin/bash
# tools/ab_stream.sh <tag> — streamed scenes (10k and 100k spheres, BASELINE configs[3]): the matrix-pipe filter in word-local
# frames against the staged vector-ALU walk (RTK_NO_LOCAL_FRAMES=1), then the beyond-LDS parity tests
TAG=$1
mkdir -p gpurun_out
for nolocal in 1 0; do
  for k in 50 158; do
    spp=16; [ $k = 158 ] && spp=2
    if [ $nolocal = 1 ]; then export RTK_NO_LOCAL_FRAMES=1; else unset RTK_NO_LOCAL_FRAMES; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --scene-k $k --spp $spp > gpurun_out/${TAG}_k${k}_nolocal$nolocal.json 2> gpurun_out/${TAG}_k$k.err || { echo FAILED $nolocal $k; tail -5 gpurun_out/${TAG}_k$k.err; exit 1; }
    python - $nolocal $k gpurun_out/${TAG}_k${k}_nolocal$nolocal.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1]); r = j["roofline"]
print("%-22s k=%-4s %8.1f Mrays/s  kernel %8.2f ms  exact tests/ray %5.2f  lane util %.3f  local words %d" % ("staged vector walk" if sys.argv[1] == "1" else "word-local matrix pipe", sys.argv[2], j["value"], r["kernel_ms"], r["exact_tests_per_ray"], r["lane_utilisation"], r["class_words"]), flush=True)
PY
  done
done
unset RTK_NO_LOCAL_FRAMES
timeout -k 10 400 python -u -m pytest tests/test_gpu_y_config3.py tests/test_gpu_intersect.py -x -q -m gpu --timeout 120 --timeout-method thread 2>&1 | tail -3
