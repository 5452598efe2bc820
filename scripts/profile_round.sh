#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats of the default bench command + PMC passes for the
# dominant kernel.  Usage: scripts/profile_round.sh <tag> [precision]
TAG=${1:-r01}; PREC=${2:-f16x3}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$PREC -o t -- python $R/bench.py --steps 50 --warmup 10 --precision $PREC --no-cpu-baseline --no-other-modes > $OUT/bench_under_rocprof_$PREC.json 2> $OUT/trace_$PREC.err
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES"; do
  n=$(echo $set | cut -d" " -f1)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_${PREC}_$n -o p -- python $R/bench.py --steps 5 --warmup 2 --precision $PREC --no-cpu-baseline --no-other-modes > /dev/null 2> $OUT/pmc_${PREC}_$n.err
done
ls -R $OUT | head -40
