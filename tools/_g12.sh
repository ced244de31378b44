set -e
mkdir -p gpurun_out/r3b
timeout -k 10 300 python -u -m pytest tests/test_gpu_train.py tests/test_layers.py -x -q --timeout 180 --timeout-method thread > gpurun_out/r3b/t.log 2>&1
timeout -k 10 200 python bench.py --workload train_dagnet_bsk_8192 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3b/tr.log 2>&1
