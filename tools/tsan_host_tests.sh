#!/bin/bash
# ThreadSanitizer run of the HOST code of libclsimhip.so (CPU only: GPU AddressSanitizer is not available on the pool).
# Builds a copy of the library whose host side is instrumented (-Xarch_host -fsanitize=thread) in /tmp and runs the
# CPU tests of the host logic (feeder threads and queues, step store, light-source front end, wire format, table
# compiler, flasher planning) against it.  ANALYSIS TOOL.   usage: tools/tsan_host_tests.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/clsimhip_tsan
rm -rf $W && mkdir -p $W && cp -r $ROOT/clsim_amd $ROOT/oracle $ROOT/tests $ROOT/include $ROOT/__graft_entry__.py $W/
make -s -C $W/clsim_amd/csrc clean
make -s -j8 -C $W/clsim_amd/csrc EXTRA="-Xarch_host -fsanitize=thread -Xarch_host -fno-omit-frame-pointer -Xarch_host -g" 2>&1 | grep -v "argument unused" || true
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.tsan-x86_64.so)
cd $W
LD_PRELOAD=$RT TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4 log_path=/tmp/clsimhip_tsan/tsan" python -m pytest tests/test_feeder.py tests/test_step_store.py tests/test_lightsource.py tests/test_abi.py -x -q -m "not gpu" -p no:cacheprovider
