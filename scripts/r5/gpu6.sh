#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
( time timeout 900 python bench.py --no-cpu-baseline --no-other-modes --no-train-key ) > gpurun_out/r5/bench_live_traffic.json 2> gpurun_out/r5/bench_live_traffic.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5/bench_live_traffic.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['roofline']['traffic'], d['roofline'].get('traffic_source'))
PY
tail -4 gpurun_out/r5/bench_live_traffic.err
( time timeout 900 python bench.py --mode train --no-cpu-baseline ) 2>&1 | python -c "
import sys,json
t=sys.stdin.read(); l=[x for x in t.splitlines() if x.startswith('{')][-1]; d=json.loads(l); print('train', d['ms_per_step'], d['roofline']['traffic'], d['roofline'].get('traffic_source','')[:80]); print(t[-120:])"
