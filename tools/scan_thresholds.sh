#!/bin/bash
# tools/scan_thresholds.sh WORKLOAD OUT: the launcher's thresholds one at a time around their automatic values on one bench.py workload (c2 | c3 | c5 |
# benchmark), developer build (tools/build_variant.sh dev).  GPU box, through gpurun, from the repo root; photons/s per value into OUT.
w=${1:-c2}; out=${2:-gpurun_out/scan_thresholds_$w.txt}
mkdir -p "$(dirname "$out")"
{
  for i in 1 2 3; do tools/scan_env.sh CLSIMHIP_NOTHING 0 -- --workload $w; done
  tools/scan_env.sh CLSIMHIP_K_POP 3 4 5 6 8 -- --workload $w
  tools/scan_env.sh CLSIMHIP_K_SEARCH 1 3 5 8 -- --workload $w
  tools/scan_env.sh CLSIMHIP_K_WAIT 4 8 16 32 -- --workload $w
  tools/scan_env.sh CLSIMHIP_K_AIM 4 8 12 16 -- --workload $w
  tools/scan_env.sh CLSIMHIP_SLICES 8 12 16 24 32 -- --workload $w
  tools/scan_env.sh CLSIMHIP_K_NEW 8 16 24 32 48 -- --workload $w
} > "$out" 2>&1
