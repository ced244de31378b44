# A/B of the training step against another build of the library: bash tools/ab_lib.sh <other.so>
set -e
OLD=$(realpath $1)
timeout -k 10 400 python -u -m pytest tests/test_gpu_train.py tests/test_layers.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
bash tools/bench_train.sh new1
bash tools/bench_train.sh old1 DAGNET_HIP_LIB=$OLD
bash tools/bench_train.sh new2
bash tools/bench_train.sh old2 DAGNET_HIP_LIB=$OLD
