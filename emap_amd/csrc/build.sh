#!/bin/bash
# Build libemap_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=${EMAP_OUT:-../lib}
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${EMAP_HIPCC_FLAGS}"
pids=()
for f in udf_mlp udf_mlp_bf16 udf_mlp_bf16x3 udf_mlp_f16 udf_mlp_f16x3 sampler api; do
  $HIPCC $FLAGS -c $f.hip -o $OUT/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/libemap_hip.so $OUT/udf_mlp.o $OUT/udf_mlp_bf16.o $OUT/udf_mlp_bf16x3.o $OUT/udf_mlp_f16.o $OUT/udf_mlp_f16x3.o $OUT/sampler.o $OUT/api.o
echo "built $OUT/libemap_hip.so"
