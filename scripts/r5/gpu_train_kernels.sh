#!/bin/bash
# every kernel of the training step under rocprofv3 --kernel-trace --stats (eager launches, 40 steps) + the bench lines (train graph / eager, render)
cd "$(dirname "$0")/../.."
R=$PWD; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
d=/tmp/prof_train; rm -rf $d
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python $R/bench.py --mode train --steps 40 --warmup 10 --graph off --no-cpu-baseline --no-other-modes --no-parity --no-train-key --traffic off > /dev/null 2> /tmp/prof_train.err || tail -5 /tmp/prof_train.err)
f=$(find $d -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' | tee gpurun_out/r5/train_kernels.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 50
tot = 0.0
for r in rows:
    per = float(r["TotalDurationNs"]) / steps / 1000
    tot += per
    print("%-104s calls/step %5.2f avg %8.1f us per-step %7.1f us" % (r["Name"][:104], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1000, per))
print("sum per step %.1f us" % tot)
PY
for g in on off; do python bench.py --mode train --steps 60 --warmup 10 --graph $g --no-cpu-baseline --no-other-modes --no-parity --traffic static 2>/dev/null | tail -1 > gpurun_out/r5/bench_train_$g.json; python -c "
import json; d=json.load(open('gpurun_out/r5/bench_train_$g.json')); print('train graph=$g', d['ms_per_step'], d.get('backward_kernels'))"; done
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --traffic static 2>/dev/null | tail -1 > gpurun_out/r5/bench_render.json; python -c "
import json; d=json.load(open('gpurun_out/r5/bench_render.json')); print('render', d['ms_per_step'], d['roofline']['frac'], d.get('train', {}).get('ms_per_step'), {k: v.get('ms_per_step') for k, v in d.items() if isinstance(v, dict) and 'dropin' in k})"
