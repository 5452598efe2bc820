#!/bin/bash
# average duration of kernels matching <pattern> in the training step, per library variant: scripts/probes/kernel_time.sh <pattern> <variants...>
PAT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset EMAP_HIP_LIB; else export EMAP_HIP_LIB=$R/emap_amd/lib/$v/libemap_hip.so; fi
  rm -rf /tmp/kt_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -o t -- python $R/bench.py --mode train --steps 40 --warmup 10 --no-cpu-baseline --no-other-modes --no-parity > /dev/null 2>&1
  f=$(find /tmp/kt_$v -name "*kernel_stats.csv" | head -1)
  echo "$v: $(grep "$PAT" $f | awk -F, '{printf "%s calls avg %.1f us; ", $(NF-6), $(NF-4)/1000}')"
done
