#!/bin/bash
# tools/gpu_session.sh <stage> [tag] — the GPU steps of a round, each under its own time limit, stopping at the first
# failure (nothing more is started on the GPU after a fault, abort or time-out).  Run through gpurun, e.g.
#   gpurun --timeout 900 -- 'bash tools/gpu_session.sh verify r02'
# Build everything on the CPU first: python -m raytracinginoneweekend_amd.build && python tools/build_variants.py <specs> &&
#   hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o tools/bin/valu_rates
# Stages:
#   smoke    __graft_entry__.smoke(): one small invocation against the oracle     -> gpurun_out/<tag>_smoke.log
#   verify   GPU tests, then the headline bench line                       -> gpurun_out/<tag>_tests.log, <tag>_bench.json
#   rates    issue rates of candidate filter instructions                  -> gpurun_out/<tag>_valu_rates.txt
#   variants parity tests of each named experiment build, then their interleaved A/B -> gpurun_out/<tag>_tests_<name>.log, <tag>_ab_*.txt
#   profile  rocprofv3 kernel trace + PMC passes (tools/profile.sh)        -> gpurun_out/prof_<tag>/
#   config4  list walk vs uniform grid on the 10k-sphere scene             -> gpurun_out/<tag>_config4_*.json
#   overlap  one vs two frames in flight, full frame and an eighth of it   -> gpurun_out/<tag>_overlap*.json
set -u
STAGE=${1:?stage}
TAG=${2:-r02}
mkdir -p gpurun_out
export TMPDIR=/tmp
step() {   # step <seconds> <log> <command...>
    local secs=$1 log=$2; shift 2
    echo "== $* (limit ${secs}s) -> $log"
    timeout -k 10 "$secs" "$@" > "$log" 2>&1
    local rc=$?
    if [ $rc -ne 0 ]; then echo "FAILED rc=$rc: $*"; tail -15 "$log"; exit 1; fi
    tail -3 "$log"
}
case "$STAGE" in
  smoke)     # the smallest possible first contact with the GPU: one intersect batch and one small frame against the oracle
    step 300 gpurun_out/${TAG}_smoke.log python -c "import __graft_entry__ as g; g.smoke()" ;;
  verify)
    step 900 gpurun_out/${TAG}_tests.log python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread
    step 400 gpurun_out/${TAG}_bench.json python bench.py ;;
  rates)
    step 120 gpurun_out/${TAG}_valu_rates.txt tools/bin/valu_rates 4 20000
    step 120 gpurun_out/${TAG}_valu_rates_1wave.txt tools/bin/valu_rates 1 20000 ;;
  variants)  # gpu_session.sh variants <tag> <spp> <name> [<name> ...]: GPU parity tests against each named build; the A/B afterwards
             # times only builds that passed (the first failure ends the call: nothing more is started on the GPU)
    SPP=${3:?spp}; shift 3
    for NAME in "$@"; do
      step 600 gpurun_out/${TAG}_tests_${NAME}.log env RTK_LIB=$PWD/raytracinginoneweekend_amd/variants/librtk_${NAME}.so python -u -m pytest tests/test_gpu_intersect.py tests/test_gpu_render.py -m gpu -x -q --timeout 120 --timeout-method thread
    done
    step 900 gpurun_out/${TAG}_ab_$(echo "$@" | tr ' ' '_').txt python tools/ab_bench.py $SPP 2 "$@" ;;
  profile)
    bash tools/profile.sh "$TAG" || exit 1 ;;
  config4)
    step 300 gpurun_out/${TAG}_config4_list.json python bench.py --scene-k 50 --width 400 --height 225 --spp 16 --steps 3 --warmup 1 --no-cpu-baseline
    step 300 gpurun_out/${TAG}_config4_grid.json python bench.py --scene-k 50 --width 400 --height 225 --spp 16 --steps 3 --warmup 1 --no-cpu-baseline --accel grid
    step 300 gpurun_out/${TAG}_stock_grid.json python bench.py --accel grid --no-cpu-baseline ;;
  overlap)
    step 300 gpurun_out/${TAG}_overlap1.json python bench.py --no-cpu-baseline
    step 300 gpurun_out/${TAG}_overlap2.json python bench.py --no-cpu-baseline --frames-in-flight 2
    step 300 gpurun_out/${TAG}_overlap2_small.json python bench.py --no-cpu-baseline --frames-in-flight 2 --width 424 --height 283 --steps 40   # one eighth of the frame: the per-GPU share at N = 8
    step 300 gpurun_out/${TAG}_overlap1_small.json python bench.py --no-cpu-baseline --width 424 --height 283 --steps 40 ;;
  *) echo "unknown stage $STAGE"; exit 2 ;;
esac
echo "stage $STAGE done"
