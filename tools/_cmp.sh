set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_gemm.py -x -q -k "gru_layer" --timeout 120 --timeout-method thread 2>&1 | tail -12
timeout -k 10 400 python -u -m pytest tests/test_gpu_train.py tests/test_gpu_loop.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -12
timeout -k 10 300 python -u bench.py --workload train_dagnet_bsk_8192 --steps 10 --warmup 3 | cut -c1-300
