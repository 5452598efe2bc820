for n in 0 4 5 6 7; do
  if [ $n = 0 ]; then L=; else L=/root/repo/emap_amd/lib/abl$n/libemap_hip.so; fi
  for pr in f16x3 bf16; do for m in grad value; do
    echo "abl=$n $(EMAP_HIP_LIB=$L python scripts/gpu_kernel_only.py $pr $m 262144 10 2>&1 | tail -1)"
  done; done
done
