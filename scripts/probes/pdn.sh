cd /tmp && export TMPDIR=/tmp
for v in "" pd3 pd4; do
  if [ -z "$v" ]; then L=; else L=$GRAFT_REPO_ROOT/emap_amd/lib/$v/libemap_hip.so; fi
  for pr in f16x3 bf16; do
    rm -rf /tmp/pp; EMAP_HIP_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --precision $pr --no-cpu-baseline --no-other-modes > /dev/null 2>&1
    echo "${v:-pd2} $pr: $(grep 'false, 8' $(find /tmp/pp -name '*kernel_stats.csv' | head -1) | awk -F'","' '{print $1}' | awk -F, '{print $(NF-6), $(NF-4), $(NF-3)}' | head -1) | $(grep 'false, 8' $(find /tmp/pp -name '*kernel_stats.csv' | head -1) | cut -d, -f4-6)"
  done
done
