#!/bin/bash
# A/B on the GPU box: the value kernels with and without LASTH (EMAP_FS2_FUSE_LAST), interleaved rounds in separate processes.
# emap_amd/lib/nofs = scripts/build_variant.sh nofs -DEMAP_FS2_FUSE_LAST=0
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for round in 1 2 3; do
  for lib in emap_amd/lib/libemap_hip.so emap_amd/lib/nofs/libemap_hip.so; do
    echo "$lib: render 512 x 128 (ms per step, median, rev32 us)"
    EMAP_HIP_LIB=$R/$lib python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-modes --no-parity --no-train-key --traffic off | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median'], d['roofline']['avg_launch_us'], d['roofline']['shader_clock_mhz'])"
  done
done
for lib in emap_amd/lib/libemap_hip.so emap_amd/lib/nofs/libemap_hip.so; do
  for P in 8192 32768; do
    echo -n "$lib value pass: "
    EMAP_HIP_LIB=$R/$lib python scripts/gpu_kernel_only.py f16x3 value $P 300
  done
done
