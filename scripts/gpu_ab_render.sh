#!/bin/bash
# Same-box A/B of library variants on the forward render: scripts/gpu_ab_render.sh <out.jsonl> <variants...> ("base" = shipped)
OUT=$1; shift
: > "$OUT"
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset EMAP_HIP_LIB; else export EMAP_HIP_LIB=$PWD/emap_amd/lib/$v/libemap_hip.so; fi
    line=$(python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-modes --no-parity --no-train-key --traffic static 2>/dev/null | tail -1)
    echo "{\"variant\": \"$v\", \"round\": $round, \"line\": $line}" >> "$OUT"
  done
done
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); b = d["line"]
    print(d["variant"], d["round"], "ms/step %.4f" % b["ms_per_step"], "kernel us %.1f" % b["roofline"]["avg_launch_us"], "frac %.4f" % b["roofline"]["frac"], "clock MHz", b["roofline"].get("shader_clock_mhz"))
PY
