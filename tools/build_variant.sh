#!/bin/bash
# tools/build_variant.sh NAME [extra compiler flags]: builds the working tree into build_variants/NAME.so (for tools/ab_bench.py);
# clsim_amd/libclsimhip.so is rebuilt without the flags afterwards.  Variants are DEVELOPER builds (-DCLSIMHIP_DEVELOPER): they honour
# the tuning environment variables the scan tools set (CLSIMHIP_K_POP, CLSIMHIP_GRID, CLSIMHIP_KERNEL ...); the default build does not.
# `tools/build_variant.sh dev` is the plain developer build the scan tools look for (CLSIMHIP_LIB=build_variants/dev.so).
# Experiments that did not ship live as patches under tools/experiments/: `PATCH=tools/experiments/x.patch tools/build_variant.sh x`
# applies the patch for the variant's build and takes it out again.
set -e
cd "$(dirname "$0")/../clsim_amd/csrc"
name=$1; shift
if [ -n "$PATCH" ]; then (cd ../.. && git apply "$PATCH") || exit 1; fi
if [ -n "$POOL_ONLY" ]; then touch prop_pool_kernel.hip; else touch prop_pool_kernel.hip prop_kernel.hip; fi
touch converter.cpp tabulator.cpp feeder.cpp
make -j8 DEVELOPER=1 EXTRA="$*" 2>&1 | grep -E "error|warning: (variable|unused)" || true
mkdir -p ../../build_variants
cp ../libclsimhip.so ../../build_variants/$name.so
echo "built build_variants/$name.so with: $*"
# leave the default build behind, not the variant
if [ -n "$PATCH" ]; then (cd ../.. && git apply -R "$PATCH"); fi
touch converter.cpp tabulator.cpp feeder.cpp
if [ -n "$POOL_ONLY" ]; then touch prop_pool_kernel.hip; else touch prop_pool_kernel.hip prop_kernel.hip; fi
make -j8 2>&1 | grep -E "error" || true
