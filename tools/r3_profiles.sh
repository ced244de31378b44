#!/bin/bash
# Round 3 evidence: kernel trace + stats of every bench workload, counter passes of the default and the SDD-shaped one, shader
# clock under load, world-1 RCCL runs.  (GPU box, via gpurun; results under gpurun_out/r3prof, copied to profiles/ afterwards)
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r3prof
bash tools/profile_round.sh r03 dagnet_bsk_4096x20
cp $(find gpurun_out/prof_r03/trace -name t_kernel_stats.csv | head -1) gpurun_out/r3prof/r03_bench_kernel_stats.csv
python3 tools/pmc_summary.py gpurun_out/prof_r03 r03 dagnet_bsk_4096x20
bash tools/profile_round.sh r03sdd dagnet_sdd_4096x20
python3 tools/pmc_summary.py gpurun_out/prof_r03sdd r03 dagnet_sdd_4096x20
cp $(find gpurun_out/prof_r03sdd/trace -name t_kernel_stats.csv | head -1) gpurun_out/r3prof/r03_dagnet_sdd_4096x20_kernel_stats.csv
cp profiles/r03_pmc.json gpurun_out/r3prof/
for wl in avrnn_bsk_512x20 avrnn_bsk_4096x20 vrnn_bsk_4096x20 train_dagnet_bsk_8192; do
  OUT=gpurun_out/r3prof/trace_$wl; mkdir -p $OUT
  EXTRA="--no-reference-loop"; [ "$wl" = train_dagnet_bsk_8192 ] && EXTRA=""
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $EXTRA --workload $wl > $OUT/run.log 2>&1
  cp $(find $OUT -name t_kernel_stats.csv | head -1) gpurun_out/r3prof/r03_${wl}_kernel_stats.csv
  grep '^{"metric' $OUT/run.log | cut -c1-200
done
bash tools/clock_under_load.sh gpurun_out/r3prof/clk > gpurun_out/r3prof/clock.txt 2>&1
cat gpurun_out/r3prof/clock.txt
bash tools/r3_dist.sh gpurun_out/r3prof/dist
