#!/bin/bash
# five-axis table maker: workgroups per CU (its per-trip pool takes 4 KB of LDS per wave)
run() { python3 bench.py --workload tab5 --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c 'import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.4g photons/s  %.1f ms" % (r["value"], r["kernel_ms_per_pass"]))'; }
for g in 768 1024 1280; do echo "n=262144 grid $g: $(CLSIMHIP_GRID=$g run)"; done
for g in 768 1024 1280; do echo "n=524288 grid $g: $(CLSIMHIP_GRID=$g run --bunch 524288)"; done
echo "n=524288 default: $(run --bunch 524288)"
