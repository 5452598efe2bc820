#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_multi.py -x -q -k "oneshot" 2>&1 | tail -25 | tee gpurun_out/r5/t4.log
timeout 300 python bench.py --gpus 2 --backend gloo --mode train --allreduce oneshot --eikonal-sync local --steps 20 --warmup 5 --settle-steps 5 --no-cpu-baseline --no-other-modes --no-parity 2>&1 | tail -1 | cut -c1-1200
