#!/bin/bash
# GPU box: the wide goal head's tests, then the SDD-shaped bench line with fp16x2 (default) and bf16x3 wide layers.
set -e
OUT=${1:-gpurun_out/sdd}; mkdir -p $OUT
timeout -k 10 300 python -u -m pytest tests/test_gpu_gemm.py tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -x -q -s -k "wide or g960 or sdd" --timeout 120 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
grep -E "passed|failed|max error" $OUT/pytest.log
for tag in h2 bf3; do
  [ $tag = bf3 ] && export DAGNET_WIDE_BF16X3=1
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-reference-loop --workload dagnet_sdd_4096x20 --steps 3 --warmup 1 > $OUT/$tag.json 2> $OUT/$tag.err || { tail -5 $OUT/$tag.err; exit 1; }
  python - <<PY
import json
d = json.load(open("$OUT/$tag.json"))
print("$tag", "%.3f M traj/s" % (d["value"] / 1e6), "ms %.2f" % d["roofline"]["kernel_ms"], "ade %.6f fde %.6f" % (d["ade"], d["fde"]), d["dtype"])
PY
done
