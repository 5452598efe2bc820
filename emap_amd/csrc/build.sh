#!/bin/bash
# Build libemap_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=${EMAP_OUT:-../lib}
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${EMAP_HIPCC_FLAGS}"
pids=()
mkdir -p "$OUT/isa"
for f in udf_mlp udf_mlp_bf16 udf_mlp_bf16x3 udf_mlp_f16 udf_mlp_f16x3 sampler extraction wgrad rays train allreduce api; do
  # -save-temps keeps the gfx950 assembly of each unit: isa_lint.py checks the inline-asm load pipelines in it
  ( cd "$OUT/isa" && $HIPCC $FLAGS -save-temps -c "$OLDPWD/$f.hip" -o ../$f.o 2> $f.log || { cat $f.log; exit 1; } ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/libemap_hip.so $OUT/udf_mlp.o $OUT/udf_mlp_bf16.o $OUT/udf_mlp_bf16x3.o $OUT/udf_mlp_f16.o $OUT/udf_mlp_f16x3.o $OUT/sampler.o $OUT/extraction.o $OUT/wgrad.o $OUT/rays.o $OUT/train.o $OUT/allreduce.o $OUT/api.o
python3 ../../scripts/isa_lint.py $OUT/isa/udf_mlp_*gfx950*.s $OUT/isa/wgrad*gfx950*.s
find $OUT/isa -type f ! -name "*gfx950*.s" -delete   # keep only the device assembly
echo "built $OUT/libemap_hip.so"
