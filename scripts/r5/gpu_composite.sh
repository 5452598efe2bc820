#!/bin/bash
# register-resident composite / composite_bwd kernels: parity tests, then rocprofv3 kernel durations of the per-ray kernels at 512 and 4096 rays
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5
[ "$1" = notest ] || timeout 900 python -m pytest tests -m gpu -x -q -k "composite or render or golden or dropin or trainer" 2>&1 | tail -4
R=$PWD
export TMPDIR=/tmp
for rays in 512 4096; do
  d=/tmp/prof_comp_$rays; rm -rf $d
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python $R/bench.py --mode train --rays $rays --steps 20 --warmup 3 --graph off --no-cpu-baseline --no-other-modes --no-parity --no-train-key --traffic off > /dev/null 2> /tmp/prof_comp.err || tail -5 /tmp/prof_comp.err)
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $rays rays"; python - "$f" <<'PY' | tee -a gpurun_out/r5/composite_kernels.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("composite", "sampler_step")):
        print("%-110s calls %4s avg %9.1f ns" % (n[:110], r["Calls"], float(r["AverageNs"])))
PY
done
