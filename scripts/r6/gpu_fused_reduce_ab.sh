#!/bin/bash
# A/B on the GPU box: the cross-ray reduction inside the value + grad_x launch (3 launches per render) against composite_reduce_kernel as a
# fourth launch (emap_amd/lib/nored = EMAP_VARIANT_UNITS=api scripts/build_variant.sh nored -DEMAP_FUSED_REDUCE=0); interleaved, separate processes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for round in 1 2 3 4; do
  for lib in emap_amd/lib/libemap_hip.so emap_amd/lib/nored/libemap_hip.so; do
    echo -n "$lib: render 512 x 128 (ms per step, median, rev32 us, MHz): "
    EMAP_HIP_LIB=$R/$lib python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-modes --no-parity --no-train-key --traffic off | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median'], d['roofline']['avg_launch_us'], d['roofline']['shader_clock_mhz'])"
  done
done
