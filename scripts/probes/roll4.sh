cd /tmp && export TMPDIR=/tmp
for v in "" roll4; do
  if [ -z "$v" ]; then L=; else L=$GRAFT_REPO_ROOT/emap_amd/lib/$v/libemap_hip.so; fi
  for pr in f16x3 bf16; do
    rm -rf /tmp/pp; EMAP_HIP_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --precision $pr --no-cpu-baseline --no-other-modes > /dev/null 2>&1
    echo "${v:-base} $pr: $(grep 'false, 4' $(find /tmp/pp -name '*kernel_stats.csv' | head -1) | cut -d, -f5-7)"
  done
done
cd $GRAFT_REPO_ROOT; EMAP_HIP_LIB=$GRAFT_REPO_ROOT/emap_amd/lib/roll4/libemap_hip.so python -m pytest tests -q -m gpu 2>&1 | tail -1
