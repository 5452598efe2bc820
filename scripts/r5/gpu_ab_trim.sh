#!/bin/bash
# same-box A/B of the MX K-loop's loads in flight (EMAP_REV_MXD_TRIM): correctness first, then the forward step interleaved
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5
for v in "$@"; do
  if [ "$v" = base ]; then unset EMAP_HIP_LIB; else export EMAP_HIP_LIB=$PWD/emap_amd/lib/$v/libemap_hip.so; fi
  echo "== $v"; python scripts/gpu_ab_rev.py f16x3 --variants rev --rounds 5 --no-oracle 2>/dev/null | tail -2 | cut -c1-400
done
unset EMAP_HIP_LIB
bash scripts/gpu_ab_render.sh gpurun_out/r5/ab_trim.jsonl "$@" 2>/dev/null | tail -12
