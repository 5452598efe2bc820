#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in base nosmx; do
  if [ "$v" = base ]; then unset EMAP_HIP_LIB; else export EMAP_HIP_LIB=$PWD/emap_amd/lib/$v/libemap_hip.so; fi
  echo "== $v"; python -m pytest tests/test_gpu_backward.py -q -s -k "golden or mirror or trajectory or convergence" 2>&1 | grep -iE "worst|error|err |rel|passed|failed|cos" | head -24
done > gpurun_out/r5/smx_errors.txt 2>&1
unset EMAP_HIP_LIB
cat gpurun_out/r5/smx_errors.txt
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r5/full_gpu_tests2.log
