#!/bin/bash
# tools/scan_small_shard.sh: C5's per-GPU shards (fewer steps than the chip has unit slots) on the classic and the pooled kernel with
# rings of several sizes (developer build / round-5 environment interface).  photons/s per setting.
# needs a developer build: tools/build_variant.sh dev (the default library reads no tuning from the environment)
export CLSIMHIP_LIB=${CLSIMHIP_LIB:-$(dirname "$0")/../build_variants/dev.so}
[ -f "$CLSIMHIP_LIB" ] || { echo "no $CLSIMHIP_LIB: run tools/build_variant.sh dev" >&2; exit 1; }
for shard in 312500 625000; do
  for kr in classic:0 pool:4 pool:8 pool:12 pool:16 pool:24 pool:32 pool:45; do
    k=${kr%%:*}; r=${kr##*:}
    v=$(env CLSIMHIP_KERNEL=$k CLSIMHIP_POOL_R=$r python3 bench.py --workload c5 --shard-steps $shard --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-host-path | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("%.4g photons/s  kernel %.2f ms" % (d["value"], d["roofline"]["avg_kernel_ms"]))')
    echo "c5 shard $shard kernel=$k ring=$r: $v"
  done
done
