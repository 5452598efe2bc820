#!/bin/bash
# pack_all_kernel by section (variants pk<mask>; "base" = all sections)
cd "$(dirname "$0")/../.."
R=$PWD; export TMPDIR=/tmp
for v in base pk1 pk2 pk4 pk8 pk32; do
  if [ "$v" = base ]; then unset EMAP_HIP_LIB; else export EMAP_HIP_LIB=$R/emap_amd/lib/$v/libemap_hip.so; fi
  d=/tmp/prof_pack_$v; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python $R/scripts/r5/pack_time.py > /dev/null 2> /tmp/pk.err || tail -3 /tmp/pk.err)
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python - "$f" $v <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "pack_all" in r["Name"] or "rowscale" in r["Name"]:
        print(sys.argv[2], r["Name"][:40], "calls", r["Calls"], "avg %.1f us" % (float(r["AverageNs"]) / 1000))
PY
done
