#!/bin/bash
# Marginal cost of scalar and of vector instructions in the pooled propagation kernel (GPU box, from the repo root).
# ANALYSIS TOOL: rebuilds the library with N dummy instructions per loop trip and times the headline bunch.  The dummy instructions
# are tools/experiments/issue_cost.patch (applied here, taken out at the end; needs git on the box or a pre-patched snapshot).
git apply tools/experiments/issue_cost.patch 2>/dev/null || patch -p1 -s < tools/experiments/issue_cost.patch || exit 1
for flags in "" "-DCLSIMHIP_EXP_SALU=50" "-DCLSIMHIP_EXP_SALU=100" "-DCLSIMHIP_EXP_VALU=50" "-DCLSIMHIP_EXP_VALU=100"; do
  rm -f clsim_amd/csrc/prop_pool_kernel.o
  make -s -C clsim_amd/csrc EXTRA="$flags" 2>/dev/null
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('flags=[$flags]', '%.4g photons/s'%d['value'], '%.2f ms'%d['roofline']['avg_kernel_ms'])"
done
# leave the default library behind: the Makefile does not track EXTRA, so the last variant would otherwise stay in place
git apply -R tools/experiments/issue_cost.patch 2>/dev/null || patch -p1 -R -s < tools/experiments/issue_cost.patch
rm -f clsim_amd/csrc/prop_pool_kernel.o
make -s -C clsim_amd/csrc
