#!/bin/bash
# tools/gpu_determinism.sh <tag> — the determinism checks first (12 frames of the stock scene = class-word loop, 6 of the
# mixed-radii scene = general-word loop, 6 of the 10 001-sphere scene = word-local walk, each also against the uniform-grid kernel's
# frame), then the GPU tests and the
# headline bench (tools/gpu_session.sh verify)
TAG=${1:-r03}
mkdir -p gpurun_out
timeout -k 10 200 python tools/determinism_check.py 12 --grid > gpurun_out/${TAG}_determinism.txt 2>&1 || { tail -3 gpurun_out/${TAG}_determinism.txt; exit 1; }
tail -1 gpurun_out/${TAG}_determinism.txt
timeout -k 10 200 python tools/determinism_check.py 6 --grid --scene mixed > gpurun_out/${TAG}_determinism_mixed.txt 2>&1 || { tail -3 gpurun_out/${TAG}_determinism_mixed.txt; exit 1; }
tail -1 gpurun_out/${TAG}_determinism_mixed.txt
timeout -k 10 300 python tools/determinism_check.py 6 --grid --scene scaled > gpurun_out/${TAG}_determinism_scaled.txt 2>&1 || { tail -3 gpurun_out/${TAG}_determinism_scaled.txt; exit 1; }
tail -1 gpurun_out/${TAG}_determinism_scaled.txt
bash tools/gpu_session.sh verify ${TAG}
