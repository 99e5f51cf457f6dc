#!/bin/bash
# tools/r03_scene_survey.sh <tag> — one bench line per scene kind at 1200x800 (spp 100 unless noted), list walk:
# stock, configs[4] glass at depth 100, class-word scenes of growing extent (k = 13, 15), mixed-radii scenes of 200 / 500 / 1000 spheres
TAG=${1:-r03}
ONLY=${2:-}
mkdir -p gpurun_out
run() { local name=$1; shift; if [ -n "$ONLY" ] && [[ "$name" != $ONLY* ]]; then return; fi; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 "$@" > gpurun_out/${TAG}_scene_${name}.json 2> gpurun_out/${TAG}_scene_${name}.err || { echo "FAILED $name"; tail -5 gpurun_out/${TAG}_scene_${name}.err; exit 1; }
  python - "$name" gpurun_out/${TAG}_scene_${name}.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1]); r = j["roofline"]
print("%-12s %8.0f Mrays/s  kernel %7.2f ms  exact tests/ray %5.2f  lane util %s  rays/sample %.3f  LDS/wg %6s  class words %3s  frac %s" % (sys.argv[1], j["value"], r["kernel_ms"], r["exact_tests_per_ray"], ("%.3f" % r["lane_utilisation"]) if "lane_utilisation" in r else "  -  ", j["config"]["rays_per_sample"], r.get("lds_bytes_per_workgroup", "-"), r.get("class_words", "-"), r.get("frac")))
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
# scenes beyond LDS (BASELINE configs[3]): word-local matrix-pipe walk; the same with the uniform grid
run k50 --scene-k 50 --spp 16
run k50_grid --scene-k 50 --spp 16 --accel grid
run k158 --scene-k 158 --spp 16
run k158_grid --scene-k 158 --spp 16 --accel grid
