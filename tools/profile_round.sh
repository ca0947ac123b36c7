#!/bin/bash
# Profile passes for one round (run on the GPU box through gpurun, from the repo root):
#   tools/profile_round.sh <tag>
# 1. rocprofv3 --kernel-trace --stats of the default bench command (kernel durations),
# 2. separate --pmc passes (never combined with the trace domains): issue/occupancy counters,
#    FETCH_SIZE, WRITE_SIZE (one counter per pass, as the MI355X guide prescribes for HBM traffic),
# 3. tools/summarize_profile.py -> gpurun_out/prof_<tag>/summary.json (copied into profiles/ by hand).
set -u
TAG=${1:-round}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-table-maker"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH > $OUT/bench_kt.json 2> $OUT/kt.err; echo kt rc=$?
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc1 -- $BENCH > $OUT/bench_pmc1.json 2> $OUT/pmc1.err; echo pmc1 rc=$?
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2 -- $BENCH > /dev/null 2> $OUT/pmc2.err; echo pmc2 rc=$?
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3 -- $BENCH > /dev/null 2> $OUT/pmc3.err; echo pmc3 rc=$?
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc4 -- $BENCH > /dev/null 2> $OUT/pmc4.err; echo pmc4 rc=$?
python3 tools/summarize_profile.py $OUT > $OUT/summary.json; cat $OUT/summary.json
find $OUT -name "*_kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# raw per-dispatch counter files are large: keep only the summaries
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
