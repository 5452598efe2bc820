cd /tmp && export TMPDIR=/tmp
for n in 0 2 4 6 7; do
  if [ $n = 0 ]; then L=; else L=$GRAFT_REPO_ROOT/emap_amd/lib/abl$n/libemap_hip.so; fi
  rm -rf /tmp/pa$n
  EMAP_HIP_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa$n -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --precision f16x3 --no-cpu-baseline --no-other-modes > /dev/null 2>&1
  f=$(find /tmp/pa$n -name "*kernel_stats.csv" | head -1)
  echo "abl=$n"; grep "udf_mlp_fs2" $f | awk -F'","' '{print "   ", $1, "avg_ns", $4}' | cut -c1-120
done
