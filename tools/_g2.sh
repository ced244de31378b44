set -e
mkdir -p gpurun_out/r2m
timeout -k 10 300 python -u -m pytest tests/test_gpu_gemm.py tests/test_gpu_properties.py tests/test_gpu_parity.py -x -q -k "gemm or wide or g960 or sdd" --timeout 120 --timeout-method thread > gpurun_out/r2m/t.log 2>&1 && \
timeout -k 10 120 python tools/ubench/gemm_bench.py 122880 > gpurun_out/r2m/pipe.log 2>&1 && \
timeout -k 10 300 python bench.py --workload dagnet_sdd_4096x20 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2m/sdd.log 2>&1
