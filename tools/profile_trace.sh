#!/bin/bash
# rocprofv3 kernel trace + stats of one bench workload: tools/profile_trace.sh <tag> <workload>
set -e
TAG=$1; WL=$2
OUT=gpurun_out/trace_${TAG}_$WL
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $WL > $OUT/run.log 2>&1
head -3 $OUT/t_kernel_stats.csv
