#!/bin/bash
# Round 3, VERDICT item 1(b): RCCL init -> chunked walk -> all-reduce -> destroy at world size 1 on the one-GPU box, plus the
# training step with its gradient all-reduce.  Usage (via gpurun): bash tools/r3_dist.sh <outdir>
set -e
OUT=${1:-gpurun_out/r3a}
mkdir -p $OUT
export DAGNET_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517
timeout -k 10 400 python bench.py --workload dagnet_bsk_1Mx20 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/dist_1m.json 2> $OUT/dist_1m.err
tail -c 600 $OUT/dist_1m.json; echo
timeout -k 10 300 python bench.py --workload train_dagnet_bsk_8192 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/dist_train.json 2> $OUT/dist_train.err
tail -c 600 $OUT/dist_train.json; echo
