#!/bin/bash
# final round-5 records: default bench line, train bench line, rocprofv3 kernel stats + PMC passes (render and train, f16x3), batch sweep
cd "$(dirname "$0")/../.."
R=$PWD
mkdir -p gpurun_out/r5
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/r5/bench_default.json 2> gpurun_out/r5/bench_default.err
timeout 600 python bench.py --mode train --no-cpu-baseline > gpurun_out/r5/bench_train.json 2>/dev/null
bash scripts/profile_round.sh r05 f16x3 render 512 > gpurun_out/r5/prof_render.log 2>&1
bash scripts/profile_round.sh r05 f16x3 train 512 > gpurun_out/r5/prof_train.log 2>&1
cd $R
bash scripts/batch_sweep.sh gpurun_out/r5/batch_sweep.jsonl > gpurun_out/r5/batch_sweep.log 2>&1
tail -8 gpurun_out/r5/batch_sweep.log
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5/bench_default.json') if l.startswith('{')][-1])
print('render', d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['traffic'])
print({k:(round(d[k]['ms_per_step'],3), d[k].get('vs_native_trainer')) for k in ('train','train_dropin','train_dropin_fused_adam','train_dropin_patched')}, d['train'].get('graph_replay'))
t=json.loads([l for l in open('gpurun_out/r5/bench_train.json') if l.startswith('{')][-1])
print('train', t['ms_per_step'], t['backward_kernels'], t['roofline']['frac'], t['roofline']['traffic'])
PY
