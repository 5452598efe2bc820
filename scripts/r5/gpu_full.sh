#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r5/full_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
