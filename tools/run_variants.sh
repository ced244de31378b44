#!/bin/bash
# On the GPU box: bench line of every variant library given (tools/variants.sh), default workload unless WL is set.
#   bash tools/run_variants.sh <outdir> <tag> [<tag> ...]
OUT=$1; shift
mkdir -p $OUT
for TAG in "$@"; do
  LIBF=dagnet_amd/_lib/libdagnet_hip_$TAG.so
  [ "$TAG" = base ] && LIBF=dagnet_amd/_lib/libdagnet_hip.so
  for wl in ${WL:-dagnet_bsk_4096x20}; do
    DAGNET_HIP_LIB=$PWD/$LIBF timeout -k 10 200 python bench.py --no-cpu-baseline --no-reference-loop --workload $wl > $OUT/${TAG}_$wl.json 2> $OUT/${TAG}_$wl.err || { tail -5 $OUT/${TAG}_$wl.err; exit 1; }
    python - <<PY
import json
d = json.load(open("$OUT/${TAG}_$wl.json"))
print("$TAG $wl", "%.3f M traj/s" % (d["value"] / 1e6), "kernel_ms %.2f" % d["roofline"]["kernel_ms"], "ade %.6f" % d["ade"])
PY
  done
done
