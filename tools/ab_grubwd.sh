# fused GRU backward: unit test, training parity tests, and the training step with / without it (same box)
set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_gemm.py -x -q --timeout 120 --timeout-method thread -k "gru_layer" 2>&1 | tail -5
timeout -k 10 400 python -u -m pytest tests/test_gpu_train.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -5
bash tools/bench_train.sh fused
bash tools/bench_train.sh unfused DAGNET_NO_GRU_BWD_FUSED=1
bash tools/bench_train.sh fused_w4 DAGNET_GRU_BWD_WAVES=4      # one wave per SIMD
bash tools/bench_train.sh fused2
