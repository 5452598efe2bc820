for v in "" sg4 sg8; do
  if [ -z "$v" ]; then L=; else L=/root/repo/emap_amd/lib/$v/libemap_hip.so; fi
  for pr in f16x3 bf16; do
    echo "lead=${v:-2} $(EMAP_HIP_LIB=$L python scripts/gpu_kernel_only.py $pr grad 65536 30 2>&1 | tail -1)  $(EMAP_HIP_LIB=$L python scripts/gpu_kernel_only.py $pr grad 262144 10 2>&1 | tail -1)"
  done
done
