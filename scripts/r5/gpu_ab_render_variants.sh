#!/bin/bash
# same-box A/B of library variants on the forward render: scripts/r5/gpu_ab_render_variants.sh <precision> <variant dirs under emap_amd/lib, "base" = the shipped one> ...
cd "$(dirname "$0")/../.."
PREC=$1; shift
for round in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset EMAP_HIP_LIB; else export EMAP_HIP_LIB=$PWD/emap_amd/lib/$v/libemap_hip.so; fi
    python bench.py --mode render --precision $PREC --steps 100 --warmup 10 --no-cpu-baseline --no-other-modes --no-parity --no-train-key --traffic off 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', $round, 'ms/step %.4f' % d['ms_per_step'], 'rev32 us %.1f' % d['roofline']['avg_launch_us'], 'clock', round(d['roofline'].get('shader_clock_mhz',0)))"
  done
done
