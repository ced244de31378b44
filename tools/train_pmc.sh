#!/bin/bash
# Hardware counters of the training step: one rocprofv3 --pmc pass per counter group over
#   bench.py --workload train_dagnet_bsk_8192 --no-graph   (eager: every kernel is a dispatch of its own; 1 warm-up + 3 timed steps)
# and the sum over ALL kernels, per optimizer step.  Usage (GPU box): bash tools/train_pmc.sh <outdir> [workload]
OUT=${1:-gpurun_out/train_pmc}; WL=${2:-train_dagnet_bsk_8192}
mkdir -p $OUT; export TMPDIR=/tmp
ARGS="bench.py --workload $WL --no-graph --steps 3 --warmup 1 --no-cpu-baseline"
timeout -s KILL 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o c -- python3 $ARGS > $OUT/fetch.log 2>&1 || exit 1
timeout -s KILL 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o c -- python3 $ARGS > $OUT/write.log 2>&1 || exit 1
timeout -s KILL 280 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o c -- python3 $ARGS > $OUT/sq.log 2>&1 || exit 1
python3 tools/train_pmc_summary.py $OUT $WL 4
