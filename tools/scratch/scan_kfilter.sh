#!/bin/bash
# scan of CLSIMHIP_K_FILTER / CLSIMHIP_K_FILTER_WAIT (pooled kernel, round 5) against the previous build, interleaved
OUT=gpurun_out/r5/scan_k_filter.txt
: > $OUT
run() { # lib workload kf kw
  v=$(CLSIMHIP_LIB=$1 CLSIMHIP_K_FILTER=$3 CLSIMHIP_K_FILTER_WAIT=$4 timeout -k 10 300 python3 bench.py --workload $2 --steps 6 --warmup 2 --no-cpu-baseline --no-host-path 2>/dev/null | python3 -c "import json,sys; print('%.4g' % json.loads(sys.stdin.read())['value'])")
  echo "$2 $(basename $1) k_filter=$3 wait=$4 $v" | tee -a $OUT
}
for w in c2 c3 benchmark; do
  run build_variants/r5_cells16.so $w 0 -1
  for kf in 1 2 3 4 6; do run build_variants/r5_fpark.so $w $kf 6; done
  run build_variants/r5_fpark.so $w 3 3
  run build_variants/r5_fpark.so $w 3 12
  run build_variants/r5_fpark.so $w 4 12
  run build_variants/r5_cells16.so $w 0 -1
done
run build_variants/r5_cells16.so c5 0 -1
run build_variants/r5_fpark.so c5 0 -1
