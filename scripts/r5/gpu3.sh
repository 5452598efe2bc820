#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests/test_gpu_round5.py -x -q 2>&1 | tail -15 > gpurun_out/r5/t3.log; cat gpurun_out/r5/t3.log
timeout 200 python scripts/r5/dropin_segments.py 2>/dev/null | tee gpurun_out/r5/segments_patched.txt
timeout 200 python scripts/r5/dropin_segments.py --unpatched 2>/dev/null | tee gpurun_out/r5/segments_unpatched.txt
