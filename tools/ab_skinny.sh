# A/B of skinny-kernel variants: microbenchmark and the training step, same box (experiment script)
set -e
OLD=$PWD/dagnet_amd/_lib/libdagnet_hip_skold.so
SH="64,64,0,0,1 64,64,1,0,0 64,64,1,1,0 92,64,0,0,0 90,90,1,0,0 90,90,1,1,0 64,96,1,0,0 32,64,0,0,0 32,90,1,0,0"
echo NEW; timeout -k 10 300 python tools/ubench/skinny_bench.py 81920 $SH
echo OLD; DAGNET_HIP_LIB=$OLD timeout -k 10 300 python tools/ubench/skinny_bench.py 81920 $SH
bash tools/bench_train.sh new1
bash tools/bench_train.sh old1 DAGNET_HIP_LIB=$OLD
bash tools/bench_train.sh new2
bash tools/bench_train.sh old2 DAGNET_HIP_LIB=$OLD
