#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -s -k "f16x3e" 2>&1 | tail -12 | tee gpurun_out/r5/t5a.log
timeout 900 python -m pytest tests/test_gpu_multi.py -x -q -k "stale_range or exact_lagged" 2>&1 | tail -12 | tee gpurun_out/r5/t5b.log
timeout 300 python bench.py --precision f16x3e --no-cpu-baseline --no-other-modes --no-train-key --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('f16x3e ms', d['ms_per_step'], 'kernel us', d['roofline']['avg_launch_us'], d['parity'])"
timeout 300 python bench.py --no-cpu-baseline --no-other-modes --no-train-key --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('f16x3 ms', d['ms_per_step'], 'kernel us', d['roofline']['avg_launch_us'], d['parity'])"
