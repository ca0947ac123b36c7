#!/bin/bash
# tools/scan_env.sh VAR v1 v2 ... [-- bench.py arguments]: the kernel-path bench once per value of a tuning environment variable
# (CLSIMHIP_K_POP, CLSIMHIP_K_SEARCH, CLSIMHIP_SLICES, CLSIMHIP_K_NEW, CLSIMHIP_POOL_MIN_STEPS ...); photons/s per value.
# needs a developer build: tools/build_variant.sh dev (the default library reads no tuning from the environment)
export CLSIMHIP_LIB=${CLSIMHIP_LIB:-$(dirname "$0")/../build_variants/dev.so}
[ -f "$CLSIMHIP_LIB" ] || { echo "no $CLSIMHIP_LIB: run tools/build_variant.sh dev" >&2; exit 1; }
var=$1; shift
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" == "--" ] && shift
for v in "${vals[@]}"; do
    r=$(env $var=$v python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path "$@" | grep '^{' | python3 -c 'import json,sys; print("%.4g" % json.loads(sys.stdin.readline())["value"])')
    echo "$var=$v $* $r"
done
