#!/bin/bash
# tools/r03_scene_survey.sh <tag> — one bench line per scene kind at 1200x800 (spp 100 unless noted), list walk:
# stock, configs[4] glass at depth 100, class-word scenes of growing extent (k = 13, 15), mixed-radii scenes of 200 / 500 / 1000 spheres
TAG=${1:-r03}
mkdir -p gpurun_out
run() { local name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 "$@" > gpurun_out/${TAG}_scene_${name}.json 2> gpurun_out/${TAG}_scene_${name}.err || { echo "FAILED $name"; tail -5 gpurun_out/${TAG}_scene_${name}.err; exit 1; }
  python - "$name" gpurun_out/${TAG}_scene_${name}.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1]); r = j["roofline"]
print("%-12s %8.0f Mrays/s  kernel %7.2f ms  exact tests/ray %5.2f  lane util %.3f  rays/sample %.3f  LDS/wg %6d  class words %3d  frac %s" % (sys.argv[1], j["value"], r["kernel_ms"], r["exact_tests_per_ray"], r["lane_utilisation"], j["config"]["rays_per_sample"], r["lds_bytes_per_workgroup"], r["class_words"], r["frac"]))
PY
}
run stock --spp 100
run glass --glass --depth 100 --spp 100
run glass_d50 --glass --depth 50 --spp 100
run k13 --scene-k 13 --spp 100
run k15 --scene-k 15 --spp 100
run mixed200 --mixed 200 --spp 100
run mixed500 --mixed 500 --spp 100
run mixed1000 --mixed 1000 --spp 100
