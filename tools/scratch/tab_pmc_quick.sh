#!/bin/bash
# quick instruction counters of the table maker (one --pmc pass, no trace domains)
OUT=gpurun_out/tab_pmc_quick
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc1 -- python3 bench.py --workload tab --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.txt
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/tab_pmc_quick/pmc1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "prop_kernel" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k in acc:
    print(k[:70], "launches", n[k])
    for c, v in sorted(acc[k].items()): print("   %-26s %.6g per launch" % (c, v / max(n[k], 1)))
PY
find $OUT -name "*counter_collection.csv" -delete
