#!/bin/bash
# round-5 evidence: default bench line, rocprofv3 kernel stats + PMC passes (render f16x3, train f16x3, render f16x3e), batch sweep
cd "$(dirname "$0")/../.."
R=$PWD
mkdir -p gpurun_out/r5
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python bench.py > gpurun_out/r5/bench_default.json 2> gpurun_out/r5/bench_default.err
timeout 600 python bench.py --mode train --no-cpu-baseline > gpurun_out/r5/bench_train.json 2>/dev/null
bash scripts/profile_round.sh r05 f16x3 render 512 > gpurun_out/r5/prof_render.log 2>&1
bash scripts/profile_round.sh r05 f16x3 train 512 > gpurun_out/r5/prof_train.log 2>&1
bash scripts/profile_round.sh r05 f16x3e render 512 > gpurun_out/r5/prof_render_e.log 2>&1
cd $R
bash scripts/batch_sweep.sh gpurun_out/r5/batch_sweep.jsonl > gpurun_out/r5/batch_sweep.log 2>&1
ls gpurun_out/profiles_r05 | head -30; tail -3 gpurun_out/r5/batch_sweep.log
