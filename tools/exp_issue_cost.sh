#!/bin/bash
# Marginal cost of scalar and of vector instructions in the pooled propagation kernel (GPU box, from the repo root).
# ANALYSIS TOOL: rebuilds the library with N dummy instructions per loop trip and times the headline bunch.
for flags in "" "-DCLSIMHIP_EXP_SALU=50" "-DCLSIMHIP_EXP_SALU=100" "-DCLSIMHIP_EXP_VALU=50" "-DCLSIMHIP_EXP_VALU=100"; do
  rm -f clsim_amd/csrc/prop_pool_kernel.o
  make -s -C clsim_amd/csrc EXTRA="$flags" 2>/dev/null
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('flags=[$flags]', '%.4g photons/s'%d['value'], '%.2f ms'%d['roofline']['avg_kernel_ms'])"
done
# leave the default library behind: the Makefile does not track EXTRA, so the last variant would otherwise stay in place
rm -f clsim_amd/csrc/prop_pool_kernel.o
make -s -C clsim_amd/csrc
