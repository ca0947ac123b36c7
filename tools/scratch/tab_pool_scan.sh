#!/bin/bash
# table maker: a pool of 240 entries lets four workgroups share a CU (LDS); does a fourth wave per SIMD pay now that the atomics bound the kernel?
run() { python3 bench.py --workload tab --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c 'import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.4g photons/s  %.1f ms" % (r["value"], r["kernel_ms_per_pass"]))'; }
for rep in 1 2; do
echo "pool 448, n=262144, 3 per CU: $(CLSIMHIP_LIB=$PWD/build_variants/tab_pool448.so run)"
echo "pool 240, n=262144, 3 per CU: $(CLSIMHIP_LIB=$PWD/build_variants/tab_pool240.so run)"
echo "pool 240, n=262144, 4 per CU: $(CLSIMHIP_LIB=$PWD/build_variants/tab_pool240.so CLSIMHIP_GRID=1024 run)"
done
echo "pool 448, n=524288 (3 per CU): $(CLSIMHIP_LIB=$PWD/build_variants/tab_pool448.so run --bunch 524288)"
echo "pool 240, n=524288 (4 per CU by the launcher's rule): $(CLSIMHIP_LIB=$PWD/build_variants/tab_pool240.so run --bunch 524288)"
echo "pool 240, n=524288, 3 per CU: $(CLSIMHIP_LIB=$PWD/build_variants/tab_pool240.so CLSIMHIP_GRID=768 run --bunch 524288)"
