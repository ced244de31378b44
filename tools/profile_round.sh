#!/bin/bash
# rocprofv3 evidence for one round: kernel trace + stats, then hardware counters in separate passes
# (never --pmc together with a trace domain).  Usage (on the GPU box, via gpurun): tools/profile_round.sh r01 [workload]
set -e
TAG=${1:-r01}
WORKLOAD=${2:-dagnet_bsk_4096x20}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the untimed legs (CPU baselines, the K x sample() reference loop) are skipped: they launch the same kernels at other sizes
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference-loop --workload $WORKLOAD"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ARGS > $OUT/trace.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o c -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o c -- python3 $ARGS > $OUT/pmc_write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o c -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
ls -R $OUT | head -40
