#!/bin/bash
# Same-box A/B of library variants on the training step: scripts/gpu_ab_train.sh <out.jsonl> <variant dirs under emap_amd/lib, "base" = the shipped one> ...
# two interleaved rounds; each line = bench.py --mode train (eager) with the per-step sums of the sweep and wgrad kernels
OUT=$1; shift
: > "$OUT"
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset EMAP_HIP_LIB; else export EMAP_HIP_LIB=$PWD/emap_amd/lib/$v/libemap_hip.so; fi
    line=$(python bench.py --mode train --steps 40 --warmup 10 --graph off --no-cpu-baseline --no-other-modes --no-parity --traffic static 2>/dev/null | tail -1)
    echo "{\"variant\": \"$v\", \"round\": $round, \"line\": $line}" >> "$OUT"
  done
done
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); b = d["line"]
    print(d["variant"], d["round"], "ms/step %.3f" % b["ms_per_step"], {k: (round(v, 1) if isinstance(v, float) else v) for k, v in b.get("backward_kernels", {}).items()})
PY
