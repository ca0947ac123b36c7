#!/bin/bash
# Scalar side, instruction fetch and waits of one bench workload (GPU box, through gpurun, from the repo root):
#   tools/profile_scalar.sh <tag> <bench.py arguments ...>      e.g.  tools/profile_scalar.sh r06_c2        /  r06_tab --workload tab
# Separate --pmc passes (never combined with trace domains), then tools/summarize_scalar.py -> gpurun_out/prof_<tag>/scalar_summary.json
# (SQ_INSTS_SALU / SMEM / BRANCH per wave trip, SQ_WAIT_INST_LDS, the instruction cache of the 55 KB table maker).
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-table-maker $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH > $OUT/bench_kt.json 2> $OUT/kt.err; echo kt rc=$?
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_SALU --output-format csv -d $OUT/s1 -- $BENCH > /dev/null 2> $OUT/s1.err; echo s1 rc=$?
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/s2 -- $BENCH > /dev/null 2> $OUT/s2.err; echo s2 rc=$?
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/s3 -- $BENCH > /dev/null 2> $OUT/s3.err; echo s3 rc=$?
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_TC_INST_REQ --output-format csv -d $OUT/s4 -- $BENCH > /dev/null 2> $OUT/s4.err; echo s4 rc=$?
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQC_TC_STALL --output-format csv -d $OUT/s5 -- $BENCH > /dev/null 2> $OUT/s5.err; echo s5 rc=$?
python3 tools/summarize_scalar.py $OUT > $OUT/scalar_summary.json; cat $OUT/scalar_summary.json
find $OUT -name "*_kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
