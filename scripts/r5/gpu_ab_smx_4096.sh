#!/bin/bash
cd "$(dirname "$0")/../.."
for round in 1 2; do for v in base nosmx; do
  if [ "$v" = base ]; then unset EMAP_HIP_LIB; else export EMAP_HIP_LIB=$PWD/emap_amd/lib/$v/libemap_hip.so; fi
  for rays in 1024 4096; do
    python bench.py --mode train --rays $rays --steps 30 --warmup 5 --no-cpu-baseline --no-other-modes --no-parity --traffic off 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', $rays, round(d['ms_per_step'],3), {k:round(v,1) for k,v in d['backward_kernels'].items() if k!='launches_per_step'}, d['roofline']['shader_clock_mhz'])"
  done
done; done
