#!/usr/bin/env python
"""Condense the rocprofv3 --pmc passes of tools/profile_round.sh into profiles/<tag>_pmc.json.

    python tools/pmc_summary.py gpurun_out/prof_r01b r01 dagnet_bsk_4096x20

Per rollout kernel: mean counter value per launch, HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB
(gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md "HBM"), and the MFMA pipe
busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)."""
import collections
import csv
import json
import os
import sys

src, tag, workload = sys.argv[1], sys.argv[2], sys.argv[3]
out = {"command": f"python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference-loop --workload {workload}; one rocprofv3 --pmc pass per counter group",
       "units": "FETCH_SIZE / WRITE_SIZE in KiB per launch (mean over launches)", "kernels": {}}
for name in ("pmc_fetch", "pmc_write", "pmc_sq"):
    path = os.path.join(src, name, "c_counter_collection.csv")
    if not os.path.exists(path):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    rows = [r for r in csv.DictReader(open(path)) if "rollout" in r["Kernel_Name"] or "gemm" in r["Kernel_Name"]]
    # only the full-size launches of a kernel (the bench also launches it on small batches: plan warm-up, forward_obs_ms)
    big = collections.defaultdict(int)
    for r in rows:
        big[r["Kernel_Name"]] = max(big[r["Kernel_Name"]], int(r["Grid_Size"]))
    for r in rows:
        k = r["Kernel_Name"]
        if int(r["Grid_Size"]) != big[k]:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        out["kernels"].setdefault(k.split("(")[0], {}).update({c: x / len(seen[k]) for c, x in v.items()})
for v in out["kernels"].values():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["traffic_bytes_per_launch"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
        v["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)
dst = os.path.join("profiles", f"{tag}_pmc.json")
doc = json.load(open(dst)) if os.path.exists(dst) else {}
doc[workload] = out
json.dump(doc, open(dst, "w"), indent=1)
print("wrote", dst)
