#!/bin/bash
# AddressSanitizer + UBSan over the HOST code of the library (layout tables, handle management, table export, argument
# checks): builds synference_amd/lib/libsynference_hip_san.so (make san: --cuda-host-only, kernels are launch stubs) and runs
# the CPU tests that drive it through the C ABI.  CPU only -- GPU sanitizers are not available on the pool.
set -e
cd "$(dirname "$0")/.."
make -C synference_amd/csrc -j8 san > /tmp/sf_san_build.log 2>&1 || { tail -20 /tmp/sf_san_build.log; exit 1; }
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export LD_PRELOAD=$RT
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export SYNFERENCE_HIP_LIB=$PWD/synference_amd/lib/libsynference_hip_san.so
python -m pytest -q -p no:cacheprovider tests/test_cpu_abi_and_layout.py tests/test_cpu_trainc_layout.py tests/test_cpu_nsfc_layout.py "$@"
