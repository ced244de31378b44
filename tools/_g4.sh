set -e
mkdir -p gpurun_out/r2r
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2r/prof -o t -- python3 bench.py --workload train_dagnet_bsk_8192 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2r/trp.log 2>&1
