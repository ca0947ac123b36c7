#!/usr/bin/env python3
"""Summarises tools/profile_scalar.sh: per launch of the propagation (or table-maker) kernel the scalar-side counters, the waits, the
instruction cache and the scalar data cache; derived: scalar / scalar-memory / branch instructions per 100 vector instructions and
per wave, hit rates, the share of wave cycles spent waiting (for anything / for LDS)."""
import csv, glob, json, os, sys
from collections import defaultdict
root = sys.argv[1]
out = {"kernels": {}, "counters_per_launch": {}, "launches_seen": {}, "kernel": None}
for path in glob.glob(os.path.join(root, "kt", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        out["kernels"][row["Name"]] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]), "percent": float(row["Percentage"])}
for path in glob.glob(os.path.join(root, "s[0-9]", "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: defaultdict(float))
    for row in csv.DictReader(open(path)):
        if "prop_kernel" not in row["Kernel_Name"] and "prop_pool_kernel" not in row["Kernel_Name"]:
            continue
        out["kernel"] = row["Kernel_Name"]
        acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for name, per_dispatch in acc.items():
        vals = list(per_dispatch.values())[1:] or list(per_dispatch.values())      # (the first launch is the warm-up)
        out["counters_per_launch"][name] = sum(vals) / len(vals)
        out["launches_seen"][name] = len(vals)
c = out["counters_per_launch"]
def ratio(a, b):
    return (c[a] / c[b]) if (a in c and c.get(b)) else None
d = out["derived"] = {}
d["salu_per_100_valu"] = (100 * ratio("SQ_INSTS_SALU", "SQ_INSTS_VALU")) if ratio("SQ_INSTS_SALU", "SQ_INSTS_VALU") else None
d["smem_per_100_valu"] = (100 * ratio("SQ_INSTS_SMEM", "SQ_INSTS_VALU")) if ratio("SQ_INSTS_SMEM", "SQ_INSTS_VALU") else None
d["branch_per_100_valu"] = (100 * ratio("SQ_INSTS_BRANCH", "SQ_INSTS_VALU")) if ratio("SQ_INSTS_BRANCH", "SQ_INSTS_VALU") else None
d["lds_per_100_valu"] = (100 * ratio("SQ_INSTS_LDS", "SQ_INSTS_VALU")) if ratio("SQ_INSTS_LDS", "SQ_INSTS_VALU") else None
d["vmem_per_100_valu"] = (100 * ratio("SQ_INSTS_VMEM", "SQ_INSTS_VALU")) if ratio("SQ_INSTS_VMEM", "SQ_INSTS_VALU") else None
d["wait_any_share_of_wave_cycles"] = ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES")
d["wait_inst_any_share_of_wave_cycles"] = ratio("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")
d["wait_inst_lds_share_of_wave_cycles"] = ratio("SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES")
d["active_inst_sca_share_of_busy_cycles"] = ratio("SQ_ACTIVE_INST_SCA", "SQ_BUSY_CYCLES")
d["active_inst_valu_share_of_busy_cycles"] = ratio("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES")
d["lds_bank_conflict_share_of_lds_active"] = ratio("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")
d["valu_lane_utilisation"] = (c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])) if ("SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU")) else None
d["icache_hit_rate"] = ratio("SQC_ICACHE_HITS", "SQC_ICACHE_REQ")
d["icache_misses_per_1000_valu"] = None
d["icache_miss_rate_incl_duplicates"] = ((c["SQC_ICACHE_MISSES"] + c.get("SQC_ICACHE_MISSES_DUPLICATE", 0.0)) / c["SQC_ICACHE_REQ"]) if c.get("SQC_ICACHE_REQ") and "SQC_ICACHE_MISSES" in c else None
d["scalar_dcache_hit_rate"] = ratio("SQC_DCACHE_HITS", "SQC_DCACHE_REQ")
d["scalar_dcache_miss_rate_incl_duplicates"] = ((c["SQC_DCACHE_MISSES"] + c.get("SQC_DCACHE_MISSES_DUPLICATE", 0.0)) / c["SQC_DCACHE_REQ"]) if c.get("SQC_DCACHE_REQ") and "SQC_DCACHE_MISSES" in c else None
print(json.dumps(out, indent=1))
