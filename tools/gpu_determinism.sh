#!/bin/bash
# tools/gpu_determinism.sh — the determinism check first (12 frames), then the GPU tests and the headline bench (tools/gpu_session.sh verify)
mkdir -p gpurun_out
timeout -k 10 200 python tools/determinism_check.py 12 > gpurun_out/${1:-r02}_determinism.txt 2>&1 || { tail -3 gpurun_out/${1:-r02}_determinism.txt; exit 1; }
tail -1 gpurun_out/${1:-r02}_determinism.txt
bash tools/gpu_session.sh verify ${1:-r02}
