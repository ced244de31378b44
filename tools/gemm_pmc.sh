#!/bin/bash
# GPU box: counter passes over the wide GEMM microbenchmark (one shape), DAGNET_FUSED_MASK selects the product form.
#   bash tools/gemm_pmc.sh <outdir> <mask>
OUT=$1; export DAGNET_FUSED_MASK=$2; export GEMM_ONLY=1920
mkdir -p $OUT; export TMPDIR=/tmp
run() { timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o c -- python3 tools/ubench/gemm_bench.py 122880 > $OUT/$1.log 2>&1 || { tail -5 $OUT/$1.log; exit 1; }; }
run lds "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
run sq2 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
python3 - <<PY
import csv, glob, collections
for tag in ("lds", "tcc", "sq2"):
    f = glob.glob("$OUT/%s/**/c_counter_collection.csv" % tag, recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "gemm_nt_pipe" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(tag, k)
        for c, v in d.items():
            print("   %-28s %.4g per launch (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
