#!/bin/bash
# table maker: the shape of the sector tile (CLSIMHIP_TAB_TILE = e0 e2 e3, powers of two of distance / polar angle / time bins per 64-byte sector)
for t in 111 210 120 201 102 300 030 021 012 003; do
  echo "tile $t: $(CLSIMHIP_TAB_TILE=$t python3 bench.py --workload tab --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.4g photons/s  %.1f ms  sum of weights %.10g  occupied %d" % (r["value"], r["kernel_ms_per_pass"], r["sum_of_weights_per_pass"], r["occupied_bins"]))')"
done
