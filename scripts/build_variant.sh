#!/bin/bash
# Build a variant of libemap_hip.so that differs from emap_amd/lib in some units only (extra -D flags): the units named in
# EMAP_VARIANT_UNITS (default: the f16x3 MLP unit) are compiled into emap_amd/lib/<name>/ and linked with the default build's other
# objects.  For A/B timing and debug builds (EMAP_HIP_LIB=emap_amd/lib/<name>/libemap_hip.so selects it).
#   usage: [EMAP_VARIANT_UNITS="api wgrad"] scripts/build_variant.sh <name> [hipcc flags...]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME=$1; shift
OUT=$ROOT/emap_amd/lib/$NAME
LIB=$ROOT/emap_amd/lib
UNITS=${EMAP_VARIANT_UNITS:-udf_mlp_f16x3}
mkdir -p "$OUT/isa"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
pids=()
for u in $UNITS; do
  ( cd "$OUT/isa" && $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -save-temps -c "$ROOT/emap_amd/csrc/$u.hip" -o ../$u.o 2> $u.log || { cat $u.log; exit 1; } ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
OBJS=""
for u in udf_mlp udf_mlp_bf16 udf_mlp_bf16x3 udf_mlp_f16 udf_mlp_f16x3 sampler extraction wgrad rays train allreduce api; do
  if [[ " $UNITS " == *" $u "* ]]; then OBJS="$OBJS $OUT/$u.o"; else OBJS="$OBJS $LIB/$u.o"; fi
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libemap_hip.so" $OBJS
find "$OUT/isa" -type f ! -name "*gfx950*.s" -delete
echo "built $OUT/libemap_hip.so ($UNITS: $*)"
