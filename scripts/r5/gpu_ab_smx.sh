#!/bin/bash
# same-box A/B of the MX-fp6 training sweep (default) against the three-f16-pass sweep (build nosmx: -DEMAP_SWEEP_MX=0)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5
bash scripts/gpu_ab_train.sh gpurun_out/r5/ab_smx.jsonl base nosmx 2>/dev/null | tail -8
python -m pytest tests/test_gpu_backward.py -q -s -k "reference_golden or vs_mirror" 2>&1 | grep -E "worst|max|err|passed|failed" | head -30
