#!/bin/bash
# same-box A/B of the patched drop-in step: autograd on the calling thread (default) vs on the engine's worker thread (EMAP_DROPIN_MT=1), vs the native Trainer
cd "$(dirname "$0")/../.."
run() { python bench.py --mode train --path $1 --steps 200 --warmup 30 --no-cpu-baseline --no-parity --traffic off 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', round(d['ms_per_step'],4))"; }
for round in 1 2 3; do
  run native -
  EMAP_DROPIN_MT=0 run dropin-patched single-thread-autograd
  EMAP_DROPIN_MT=1 run dropin-patched engine-thread
done
