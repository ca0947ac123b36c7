#!/bin/bash
# tools/scan_filter_params.sh LIB WORKLOADS "ENV=VALUE ..." ...: bench.py (kernel path) once per workload and setting of the search
# filter's tuning variables (CLSIMHIP_K_SEARCH / K_WAIT / K_AIM), one line each.  ANALYSIS TOOL, run through gpurun.
LIB=$1; shift; WORKLOADS=$1; shift
for W in ${WORKLOADS//,/ }; do
  for SETTING in "$@"; do
    env $SETTING CLSIMHIP_LIB=$PWD/$LIB timeout -k 10 200 python bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', '$SETTING', d['value'])"
  done
done
