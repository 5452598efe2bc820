for P in 8192 12288 16384 24576 32768; do
  echo "P=$P rev: $(EMAP_GRAD_MODE=rev python scripts/gpu_kernel_only.py f16x3 grad $P 50 | tail -1 | awk '{print $(NF-1)}')  fwd: $(EMAP_GRAD_MODE=fwd python scripts/gpu_kernel_only.py f16x3 grad $P 50 | tail -1 | awk '{print $(NF-1)}')   bf16 rev: $(EMAP_GRAD_MODE=rev python scripts/gpu_kernel_only.py bf16 grad $P 50 | tail -1 | awk '{print $(NF-1)}') fwd: $(EMAP_GRAD_MODE=fwd python scripts/gpu_kernel_only.py bf16 grad $P 50 | tail -1 | awk '{print $(NF-1)}')"
done
