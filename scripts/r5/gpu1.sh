#!/bin/bash
# round 5, GPU call 1: new tests, rev32 timeline (both precisions), element-wise A/B (MX vs no-MX reverse sweep), baseline bench
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_round5.py -x -q 2>&1 | tail -5 > gpurun_out/r5/t_round5.log
for prec in f16x3 f16x3m; do
  EMAP_HIP_LIB=$PWD/emap_amd/lib/tl/libemap_hip.so timeout 200 python scripts/probes/rev32_timeline.py --prec $prec > gpurun_out/r5/rev32_timeline_$prec.txt 2>&1
done
timeout 300 python -m pytest tests/test_gpu_round4.py -k "elementwise_relative_error_of_the_mlp" -s -q 2>&1 | grep -E "d8w|d4w|passed|failed" > gpurun_out/r5/elementwise_base.log
EMAP_HIP_LIB=$PWD/emap_amd/lib/nomx/libemap_hip.so timeout 300 python -m pytest tests/test_gpu_round4.py -k "elementwise_relative_error_of_the_mlp" -s -q 2>&1 | grep -E "d8w|d4w|passed|failed" > gpurun_out/r5/elementwise_nomx.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r5/bench_baseline.json 2> gpurun_out/r5/bench_baseline.err
tail -c 600 gpurun_out/r5/t_round5.log; cat gpurun_out/r5/elementwise_*.log; head -c 400 gpurun_out/r5/bench_baseline.json
