#!/bin/bash
# Shader clock while the rollout kernel runs: GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs) / kernel duration from the same
# rocprofv3 run (--kernel-trace with one counter).  Usage (GPU box): bash tools/clock_under_load.sh <outdir> [env assignments...]
OUT=$1; shift
mkdir -p $OUT; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT -o c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference-loop > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
tr = glob.glob("$OUT/**/c_kernel_trace.csv", recursive=True)[0]
cc = glob.glob("$OUT/**/c_counter_collection.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(tr)):
    if "rollout_v2" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"][:60])
best = max(int(r["Grid_Size"]) for r in csv.DictReader(open(cc)) if "rollout_v2" in r["Kernel_Name"])
for r in csv.DictReader(open(cc)):
    if "rollout_v2" in r["Kernel_Name"] and int(r["Grid_Size"]) == best and r["Dispatch_Id"] in dur:
        ns, name = dur[r["Dispatch_Id"]]
        cyc = float(r["Counter_Value"]) / 8
        print(f"{name}  {ns / 1e6:8.3f} ms  {cyc / 1e6:8.1f} Mcycles per XCD  -> {cyc / ns:.3f} GHz")
PY
