#!/bin/bash
# The round's kept measurements, taken in one go at the shipped revision (run on the GPU box through gpurun, from the repo root):
#   tools/final_round.sh <out dir under gpurun_out/>
# bench.py lines of every workload DESIGN.md quotes + rocprofv3 kernel stats and PMC summaries (tools/profile_round.sh /
# profile_workload.sh: --pmc passes separate, never combined with trace domains).  Copy the results into profiles/rNN/ by hand.
#   tools/final_round.sh <out dir> bench | prof     (one part per gpurun call: together they exceed a call's 20 minutes)
set -u
OUT=gpurun_out/${1:-final}
PART=${2:-all}
mkdir -p $OUT
B="timeout -k 10 400 python3 bench.py"
if [ "$PART" != "prof" ]; then
$B > $OUT/final_bench_default_line.json 2> $OUT/default.err; echo default rc=$?
$B --workload c3 --no-cpu-baseline --no-host-path --steps 5 --warmup 1 > $OUT/final_bench_c3.json 2> $OUT/c3.err; echo c3 rc=$?
$B --workload c5 --no-cpu-baseline --no-host-path --steps 5 --warmup 1 > $OUT/final_bench_c5.json 2> $OUT/c5.err; echo c5 rc=$?
$B --workload benchmark --steps 5 --warmup 1 > $OUT/final_bench_benchmark.json 2> $OUT/benchmark.err; echo benchmark rc=$?
$B --workload benchmark-host --steps 12 > $OUT/final_bench_benchmark_host.json 2> $OUT/benchmark_host.err; echo benchmark-host rc=$?
$B --workload benchmark-host --steps 4 > $OUT/final_bench_benchmark_host_8_events.json 2> $OUT/benchmark_host8.err; echo benchmark-host-8 rc=$?
$B --keep-detected --no-cpu-baseline --no-host-path --steps 10 --warmup 2 > $OUT/final_bench_c2_keep.json 2> $OUT/c2_keep.err; echo c2keep rc=$?
$B --workload c3 --keep-detected --no-cpu-baseline --no-host-path --steps 5 --warmup 1 > $OUT/final_bench_c3_keep.json 2> $OUT/c3_keep.err; echo c3keep rc=$?
$B --workload c5 --keep-detected --no-cpu-baseline --no-host-path --steps 5 --warmup 1 > $OUT/final_bench_c5_keep.json 2> $OUT/c5_keep.err; echo c5keep rc=$?
$B --workload tab --steps 3 --warmup 1 > $OUT/final_bench_tab.json 2> $OUT/tab.err; echo tab rc=$?
$B --workload tab5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/final_bench_tab5.json 2> $OUT/tab5.err; echo tab5 rc=$?
fi
if [ "$PART" = "bench" ]; then ls $OUT; exit 0; fi
timeout -k 10 500 bash tools/profile_round.sh ${1:-final}_c2 > $OUT/prof_c2.log 2>&1; echo prof c2 rc=$?
for w in c3 c5; do timeout -k 10 600 bash tools/profile_workload.sh ${1:-final}_$w --workload $w > $OUT/prof_$w.log 2>&1; echo prof $w rc=$?; done
timeout -k 10 500 bash tools/profile_workload.sh ${1:-final}_c2keep --keep-detected > $OUT/prof_c2keep.log 2>&1; echo prof c2keep rc=$?
timeout -k 10 500 bash tools/profile_workload.sh ${1:-final}_tab --workload tab > $OUT/prof_tab.log 2>&1; echo prof tab rc=$?
for w in c2 c3 c5 c2keep tab; do
  cp gpurun_out/prof_${1:-final}_$w/summary.json $OUT/final_${w}_pmc_summary.json
  cp gpurun_out/prof_${1:-final}_$w/kernel_stats.csv $OUT/final_${w}_kernel_stats.csv
  cp gpurun_out/prof_${1:-final}_$w/bench_kt.json $OUT/final_${w}_bench_under_rocprof.json
done
ls $OUT
