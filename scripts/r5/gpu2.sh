#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py -x -q -k "fused_adam or patched or direct_param or masked_adam or trainer" 2>&1 | tail -25 > gpurun_out/r5/t2.log
cat gpurun_out/r5/t2.log
for path in dropin-fused dropin-patched native; do
  timeout 300 python bench.py --mode train --path $path --steps 100 --warmup 20 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 > gpurun_out/r5/train_$path.json
  python -c "
import json,sys
d=json.load(open('gpurun_out/r5/train_$path.json')); print('$path', round(d['ms_per_step'],4), d['config'].get('host_syncs_per_step'))"
done
timeout 300 python scripts/profile_dropin_step.py --patched > gpurun_out/r5/dropin_host_profile_patched.txt 2>&1; head -30 gpurun_out/r5/dropin_host_profile_patched.txt
