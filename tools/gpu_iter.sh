#!/bin/bash
# One build->measure iteration on the GPU box: parity tests, then the bench lines of the rollout workloads.
# Usage (via gpurun): bash tools/gpu_iter.sh [tag]
set -e
TAG=${1:-iter}
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/${TAG}_tests.log 2>&1 || { tail -30 gpurun_out/${TAG}_tests.log; exit 1; }
tail -2 gpurun_out/${TAG}_tests.log
for wl in dagnet_bsk_4096x20 avrnn_bsk_512x20 vrnn_bsk_4096x20; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --workload $wl > gpurun_out/${TAG}_$wl.json 2> gpurun_out/${TAG}_$wl.err
  python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_$wl.json"))
print("$wl", "%.3f M traj/s" % (d["value"] / 1e6), "kernel_ms %.2f" % d["roofline"]["kernel_ms"], "frac %.3f" % d["roofline"]["frac"])
PY
done
