#!/bin/bash
# PMC passes for a variant library (CLSIMHIP_LIB exported by the caller) or another workload
# (BENCH_ARGS="--workload tab --steps 2 --warmup 1"): tools/profile_variant.sh <tag>
set -u
TAG=${1:-variant}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py ${BENCH_ARGS:---steps 3 --warmup 1} --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc1 -- $BENCH > $OUT/bench_pmc1.json 2> $OUT/pmc1.err; echo pmc1 rc=$?
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc4 -- $BENCH > /dev/null 2> $OUT/pmc4.err; echo pmc4 rc=$?
python3 tools/summarize_profile.py $OUT > $OUT/summary.json; cat $OUT/summary.json
find $OUT -name "*counter_collection.csv" -delete
