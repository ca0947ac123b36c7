#!/bin/bash
# Instruction-mix counters of the propagation kernel (run on the GPU box through gpurun, from the repo root):
#   tools/profile_mix.sh <tag> [bench.py arguments ...]
# Separate --pmc passes (never combined with trace domains); summary -> gpurun_out/prof_<tag>/summary.json
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path $*"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/pmc1 -- $BENCH > /dev/null 2> $OUT/pmc1.err; echo pmc1 rc=$?
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/pmc2 -- $BENCH > /dev/null 2> $OUT/pmc2.err; echo pmc2 rc=$?
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_SMEM --output-format csv -d $OUT/pmc3 -- $BENCH > /dev/null 2> $OUT/pmc3.err; echo pmc3 rc=$?
python3 tools/summarize_profile.py $OUT > $OUT/summary.json; cat $OUT/summary.json
find $OUT -name "*counter_collection.csv" -delete
