#!/bin/bash
# AddressSanitizer run of the HOST code of libclsimhip.so (CPU only: GPU AddressSanitizer is not available on the pool).
# Builds a copy of the library whose host side is instrumented (-Xarch_host -fsanitize=address) in /tmp and runs the
# CPU tests of the host logic (feeder threads and queues, step store, light-source front end, wire format, table
# compiler, flasher planning) against it.  ANALYSIS TOOL.   usage: tools/asan_host_tests.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/clsimhip_asan
rm -rf $W && mkdir -p $W && cp -r $ROOT/clsim_amd $ROOT/oracle $ROOT/tests $ROOT/include $ROOT/__graft_entry__.py $W/
make -s -C $W/clsim_amd/csrc clean
make -s -j8 -C $W/clsim_amd/csrc EXTRA="-Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -Xarch_host -g" 2>&1 | grep -v "argument unused" || true
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd $W
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 python -m pytest tests/test_feeder.py tests/test_step_store.py tests/test_lightsource.py \
    tests/test_wire_format.py tests/test_abi.py tests/test_tables.py tests/test_flasher_steps.py tests/test_stepgen.py -x -q -m "not gpu" -p no:cacheprovider
