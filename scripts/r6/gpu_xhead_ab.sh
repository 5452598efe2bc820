R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
EMAP_HIP_LIB=$R/emap_amd/lib/xh/libemap_hip.so python scripts/r6/dbg_fuse_last.py f16x3 2>&1 | grep -v amdgpu | cut -c1-160
for round in 1 2 3 4; do
  for lib in emap_amd/lib/libemap_hip.so emap_amd/lib/xh/libemap_hip.so; do
    echo -n "$lib: "
    EMAP_HIP_LIB=$R/$lib python scripts/gpu_kernel_only.py f16x3 grad 65536 300
  done
done
for round in 1 2; do
for lib in emap_amd/lib/libemap_hip.so emap_amd/lib/xh/libemap_hip.so; do
  echo -n "$lib render: "
  EMAP_HIP_LIB=$R/$lib python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-modes --no-parity --no-train-key --traffic off | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median'], d['roofline']['avg_launch_us'], d['roofline']['shader_clock_mhz'])"
done; done
