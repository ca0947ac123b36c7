#!/bin/bash
# One-GPU rates of the per-GPU shards the N > 1 lines of bench.py run (GPU box, through gpurun, from the repo root):
#   tools/make_shard_rates.sh <out dir under gpurun_out/>   ->   <out dir>/shard_*.json; tools/make_shard_rates.py folds them into
# profiles/single_gpu_shard_rates.json (the like-for-like N = 1 point of a scaling curve: same bunches, gather path on, one rank).
set -u
OUT=gpurun_out/${1:-shards}
mkdir -p $OUT
CLSIMHIP_BENCH_GATHER=1 timeout -k 10 300 python3 bench.py --shard-steps 12500000 --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $OUT/shard_c2_12500000.json 2> $OUT/c2.err; echo c2 rc=$?
for n in 312500 625000 1250000; do
  CLSIMHIP_BENCH_GATHER=1 timeout -k 10 300 python3 bench.py --workload c5 --shard-steps $n --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $OUT/shard_c5_$n.json 2> $OUT/c5_$n.err; echo c5 $n rc=$?
done
