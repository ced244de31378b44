# A/B of two builds of the skinny GEMM: microbenchmark and the training step on the same box (experiment script).
#   bash tools/ab_skinny.sh <other libdagnet_hip.so>     (e.g. one linked with gemm_skinny.o of another commit; DAGNET_HIP_LIB selects it)
set -e
OLD=$(realpath $1)
SH="64,64,0,0,1 64,64,1,0,0 64,64,1,1,0 92,64,0,0,0 90,90,1,0,0 90,90,1,1,0 64,96,1,0,0 32,64,0,0,0 32,90,1,0,0"
echo NEW; timeout -k 10 300 python tools/ubench/skinny_bench.py 81920 $SH
echo OLD; DAGNET_HIP_LIB=$OLD timeout -k 10 300 python tools/ubench/skinny_bench.py 81920 $SH
bash tools/bench_train.sh new1
bash tools/bench_train.sh old1 DAGNET_HIP_LIB=$OLD
bash tools/bench_train.sh new2
bash tools/bench_train.sh old2 DAGNET_HIP_LIB=$OLD
