#!/bin/bash
# tools/build_variant.sh NAME [extra compiler flags]: builds the working tree into build_variants/NAME.so (for tools/ab_bench.py);
# clsim_amd/libclsimhip.so is rebuilt without the flags afterwards.
set -e
cd "$(dirname "$0")/../clsim_amd/csrc"
name=$1; shift
if [ -n "$POOL_ONLY" ]; then touch prop_pool_kernel.hip; else touch prop_pool_kernel.hip prop_kernel.hip; fi
make -j8 EXTRA="$*" 2>&1 | grep -E "error|warning: (variable|unused)" || true
mkdir -p ../../build_variants
cp ../libclsimhip.so ../../build_variants/$name.so
echo "built build_variants/$name.so with: $*"
# leave the default build behind, not the variant
if [ -n "$POOL_ONLY" ]; then touch prop_pool_kernel.hip; else touch prop_pool_kernel.hip prop_kernel.hip; fi
make -j8 2>&1 | grep -E "error" || true
