for pd in 2 4 8; do
  if [ $pd = 2 ]; then L=; else L=/root/repo/emap_amd/lib/pd$pd/libemap_hip.so; fi
  for pr in f16x3 bf16; do for nct in 1 2 4; do for P in 8192 32768; do
    echo "pd=$pd nct=$nct $(EMAP_NCT=$nct EMAP_HIP_LIB=$L python scripts/gpu_kernel_only.py $pr value $P 50 2>&1 | tail -1)"
  done; done; done
done
