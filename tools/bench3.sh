#!/bin/bash
# bench lines of the three basketball rollout workloads (no tests): bash tools/bench3.sh [tag] [workloads...]
TAG=${1:-b}; shift
WLS=${@:-dagnet_bsk_4096x20 avrnn_bsk_512x20 vrnn_bsk_4096x20}
mkdir -p gpurun_out
for wl in $WLS; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --workload $wl > gpurun_out/${TAG}_$wl.json 2> gpurun_out/${TAG}_$wl.err || { tail -5 gpurun_out/${TAG}_$wl.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_$wl.json"))
print("${TAG} $wl", "%.3f M traj/s" % (d["value"] / 1e6), "kernel_ms %.2f" % d["roofline"]["kernel_ms"], "frac %.3f" % d["roofline"]["frac"])
PY
done
