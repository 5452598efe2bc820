#!/bin/bash
# Build a variant of libemap_hip.so that differs from emap_amd/lib only in the f16x3 MLP unit (extra -D flags): the unit is
# compiled into emap_amd/lib/<name>/ and linked with the default build's other objects.  For A/B timing and debug builds
# (EMAP_HIP_LIB=emap_amd/lib/<name>/libemap_hip.so selects it).   usage: scripts/build_variant.sh <name> [hipcc flags...]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME=$1; shift
OUT=$ROOT/emap_amd/lib/$NAME
LIB=$ROOT/emap_amd/lib
mkdir -p "$OUT/isa"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
( cd "$OUT/isa" && $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -save-temps -c "$ROOT/emap_amd/csrc/udf_mlp_f16x3.hip" -o ../udf_mlp_f16x3.o 2> f16x3.log || { cat f16x3.log; exit 1; } )
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libemap_hip.so" $LIB/udf_mlp.o $LIB/udf_mlp_bf16.o $LIB/udf_mlp_bf16x3.o $LIB/udf_mlp_f16.o "$OUT/udf_mlp_f16x3.o" $LIB/sampler.o $LIB/extraction.o $LIB/wgrad.o $LIB/rays.o $LIB/train.o $LIB/api.o
find "$OUT/isa" -type f ! -name "*gfx950*.s" -delete
echo "built $OUT/libemap_hip.so ($*)"
