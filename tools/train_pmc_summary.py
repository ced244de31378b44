#!/usr/bin/env python
"""Sum the counters of tools/train_pmc.sh over all kernels of the run and divide by the optimizer steps it made:
    python tools/train_pmc_summary.py <outdir> <workload> <steps incl. warm-up> [tag]
-> profiles/<tag>_pmc.json[workload]["step"] (traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB: gfx950 counts wide reads at half, MI355X_MICROARCH.md)
plus the ten kernels with the most traffic."""
import collections
import csv
import glob
import json
import os
import sys

src, wl, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
tag = sys.argv[4] if len(sys.argv) > 4 else "r03"
tot = collections.defaultdict(float)
per = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.Counter()
for sub in ("fetch", "write", "sq"):
    for f in glob.glob(os.path.join(src, sub, "**", "c_counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0][:90]
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            per[name][r["Counter_Name"]] += float(r["Counter_Value"])
            if sub == "fetch" and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                launches[name] += 1
step = {k: v / steps for k, v in tot.items()}
step["traffic_bytes_per_step"] = (2 * step.get("FETCH_SIZE", 0.0) + step.get("WRITE_SIZE", 0.0)) * 1024
step["launches_per_step"] = sum(launches.values()) / steps
if "GRBM_GUI_ACTIVE" in step:
    step["mfma_busy_frac_of_kernel_time"] = step["SQ_VALU_MFMA_BUSY_CYCLES"] / (step["GRBM_GUI_ACTIVE"] / 8 * 1024)
top = sorted(per.items(), key=lambda kv: -(2 * kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0)))[:10]
out = {"command": f"python3 bench.py --workload {wl} --no-graph --steps 3 --warmup 1 --no-cpu-baseline; one rocprofv3 --pmc pass per counter "
                  "group; sums over all kernels of the run / optimizer steps",
       "steps_in_run": steps, "step": step,
       "top_traffic_kernels": {k: {"traffic_bytes_per_step": (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 / steps,
                                   "launches_per_step": launches[k] / steps} for k, v in top}}
dst = os.path.join("profiles", f"{tag}_pmc.json")
doc = json.load(open(dst)) if os.path.exists(dst) else {}
doc[wl] = out
os.makedirs("profiles", exist_ok=True)
json.dump(doc, open(dst, "w"), indent=1)
print(json.dumps(step, indent=1))
print("wrote", dst)
