#!/bin/bash
# GPU box: L1 / L2 request counters of the default bench's rollout kernel (weight streaming).  bash tools/l2_pmc.sh <outdir>
OUT=$1; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference-loop"
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --output-format csv -d $OUT/tcc -o c -- python3 $ARGS > $OUT/tcc.log 2>&1 || { tail -3 $OUT/tcc.log; exit 1; }
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/tcp -o c -- python3 $ARGS > $OUT/tcp.log 2>&1 || { tail -3 $OUT/tcp.log; exit 1; }
python3 - <<PY
import csv, glob, collections
for tag in ("tcc", "tcp"):
    f = glob.glob("$OUT/%s/**/c_counter_collection.csv" % tag, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "rollout_v2_kernel" in r["Kernel_Name"]]
    big = max(int(r["Grid_Size"]) for r in rows)
    acc = collections.defaultdict(list)
    for r in rows:
        if int(r["Grid_Size"]) == big:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in acc.items():
        print("%-34s %.4g per launch (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
