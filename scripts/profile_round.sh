#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats of the bench command + PMC passes (separate runs, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass, and --pmc is never combined with tracing).
# Usage: scripts/profile_round.sh <tag> [precision] [mode] [rays per GPU]
TAG=${1:-r03}; PREC=${2:-f16x3}; MODE=${3:-render}; RAYS=${4:-512}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--mode $MODE --rays $RAYS --steps 50 --warmup 10 --precision $PREC --no-cpu-baseline --no-other-modes --no-parity --no-train-key --traffic off"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_${MODE}_$PREC -o t -- python $R/bench.py $ARGS > $OUT/bench_under_rocprof_${MODE}_$PREC.json 2> $OUT/trace_${MODE}_$PREC.err
PARGS="--mode $MODE --rays $RAYS --steps 5 --warmup 2 --settle-steps 5 --precision $PREC --no-cpu-baseline --no-other-modes --no-parity --no-train-key --traffic off"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL"; do
  n=$(echo $set | cut -d" " -f1)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_${MODE}_${PREC}_$n -o p -- python $R/bench.py $PARGS > /dev/null 2> $OUT/pmc_${MODE}_${PREC}_$n.err
done
python $R/scripts/summarize_profile.py $TAG $PREC $MODE > $OUT/summary_${MODE}_$PREC.txt 2>&1
tail -5 $OUT/summary_${MODE}_$PREC.txt
