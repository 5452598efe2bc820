#!/bin/bash
# bench lines at 512 / 1024 / 2048 / 4096 rays x 128 samples on ONE GPU, render and train -> <out.jsonl>   (profiles/r03_batch_sweep.jsonl)
OUT=${1:-gpurun_out/batch_sweep.jsonl}
: > "$OUT"
for mode in render train; do for rays in 512 1024 2048 4096; do
  python bench.py --mode $mode --rays $rays --steps 50 --warmup 5 --no-cpu-baseline --no-other-modes --no-parity --no-train-key --traffic static 2>/dev/null | tail -1 >> "$OUT"
done; done
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(d["config"]["mode"], d["config"]["rays_per_gpu"], "%.4g ray-samples/s" % d["value"], "%.3f ms/step" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], "clock", d["roofline"].get("shader_clock_mhz"), d.get("backward_kernels"))
PY
