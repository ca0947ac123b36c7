#!/usr/bin/env python3
"""Summarises the rocprofv3 csv output of tools/profile_round.sh: per kernel the average duration, and for the
propagation kernel the per-launch average of every collected counter."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = {"kernels": {}, "prop_kernel_counters_per_launch": {}, "launches_seen": {}}
for path in glob.glob(os.path.join(root, "kt", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        out["kernels"][row["Name"]] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]),
                                       "total_ns": float(row["TotalDurationNs"]), "percent": float(row["Percentage"])}
for path in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: defaultdict(float))
    for row in csv.DictReader(open(path)):
        if "prop_kernel" not in row["Kernel_Name"] and "prop_pool_kernel" not in row["Kernel_Name"]:
            continue
        acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for name, per_dispatch in acc.items():
        vals = list(per_dispatch.values())
        out["prop_kernel_counters_per_launch"][name] = sum(vals) / len(vals)
        out["launches_seen"][name] = len(vals)
c = out["prop_kernel_counters_per_launch"]
if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU"):
    out["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    out["hbm_bytes_per_launch_raw"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
print(json.dumps(out, indent=1))
