#!/bin/bash
# every kernel of the forward render under rocprofv3 --kernel-trace --stats + same-box render lines with the fused importance sampling on / off
cd "$(dirname "$0")/../.."
R=$PWD; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
for rays in ${RAYS:-512 1024}; do
d=/tmp/prof_render_$rays; rm -rf $d
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python $R/bench.py --mode render --rays $rays --steps 40 --warmup 10 --no-cpu-baseline --no-other-modes --no-parity --no-train-key --traffic off > /dev/null 2> /tmp/prof_render.err || tail -5 /tmp/prof_render.err)
f=$(find $d -name "*kernel_stats.csv" | head -1)
echo "== $rays rays"
python - "$f" <<'PY' | tee -a gpurun_out/r5/render_kernels.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.05:
        print("%-100s calls %5s avg %8.1f us  %s %%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1000, r["Percentage"]))
PY
for f in 1 0; do EMAP_FUSED_SAMPLING=$f python bench.py --mode render --rays $rays --steps 100 --warmup 10 --no-cpu-baseline --no-other-modes --no-parity --no-train-key --traffic off 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fused=$f rays=$rays ms/step %.4f' % d['ms_per_step'], 'value %.4g' % d['value'])"; done
done
