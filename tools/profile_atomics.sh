#!/bin/bash
# The table maker's memory-side atomics against the same counters of the micro-benchmark (GPU box, through gpurun, from the repo root):
#   tools/profile_atomics.sh <tag>
# Separate --pmc passes (never combined with trace domains) over (a) bench.py --workload tab, (b) tools/micro/atomic_rate 670 3.
# Per pass the counters of the kernel(s) of interest summed per launch -> gpurun_out/prof_<tag>/atomics_summary.json
set -u
TAG=$1
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
TAB="python3 bench.py --workload tab --steps 2 --warmup 1 --no-cpu-baseline"
[ -x tools/micro/atomic_rate ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/micro/atomic_rate tools/micro/atomic_rate.hip
MICRO="tools/micro/atomic_rate 670 3"
# (at most four counters of a block per pass: more "exceeds the capabilities of the hardware to collect", and rocprofv3 then hangs in its
# abort handler -- every pass runs under timeout)
PASSES=(
 "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_RW_ATOMIC_REQ_sum TCP_TCC_UC_ATOMIC_REQ_sum TCP_TCC_NC_ATOMIC_REQ_sum"
 "TCP_TCC_CC_ATOMIC_REQ_sum TCP_ATOMIC_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "TCC_ATOMIC_sum TCC_ATOMIC_SECTORS_sum TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum"
 "TCC_EA0_WRREQ_ATOMIC_DRAM_sum TCC_BUSY_sum TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
 "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_IB_STALL_sum"
 "TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum TCC_REQ_sum TCC_EA0_WRREQ_sum"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TA_FLAT_ATOMIC_WAVEFRONTS_sum TA_TA_BUSY_sum"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
 "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_LFIFO_STALL_CYCLES_sum"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $P --output-format csv -d $OUT/tab_p$i -- $TAB > /dev/null 2> $OUT/tab_p$i.err; echo "tab p$i rc=$? ($P)"
  timeout -k 10 150 rocprofv3 --pmc $P --output-format csv -d $OUT/micro_p$i -- $MICRO > $OUT/micro_p$i.out 2> $OUT/micro_p$i.err; echo "micro p$i rc=$?"
done
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/micro_kt -- $MICRO > $OUT/micro_kt.out 2> $OUT/micro_kt.err; echo micro kt rc=$?
python3 - $OUT <<'PY' > $OUT/atomics_summary.json
import csv, glob, json, os, sys
from collections import defaultdict
root = sys.argv[1]
out = {}
for what, match in (("tab", "prop_kernel"), ("micro", "adds")):
    per_kernel = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, what + "_p*", "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        names = {}
        for row in csv.DictReader(open(path)):
            if match not in row["Kernel_Name"]:
                continue
            acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = row["Kernel_Name"][:60]
        for c, d in acc.items():
            for disp, v in d.items():
                per_kernel[names[disp]][c].append(v)
    out[what] = {k: {c: {"launches": len(v), "first": v[0], "mean_of_the_rest": (sum(v[1:]) / len(v[1:])) if len(v) > 1 else None} for c, v in cs.items()} for k, cs in per_kernel.items()}
stats = {}
for path in glob.glob(os.path.join(root, "micro_kt", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        stats[row["Name"][:60]] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
out["micro_kernel_stats"] = stats
print(json.dumps(out, indent=1))
PY
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
head -c 3000 $OUT/atomics_summary.json
