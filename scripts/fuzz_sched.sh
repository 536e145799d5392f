#!/bin/bash
# Schedule-fuzz build of the WHOLE library (here, no GPU needed; ~6 min on 8 cores): -DSF_FUZZ_SCHED puts a wave-dependent delay behind
# every workgroup barrier (sf_device.h: __syncthreads() and the s_barrier helpers), so that a missing barrier loses its race.
# Output: synference_amd/lib/libsf_fuzz.so (git-ignored; it travels with gpurun).  Then, on the GPU box:
#   SYNFERENCE_HIP_LIB=synference_amd/lib/libsf_fuzz.so python -m pytest tests -m gpu -q
# (delete the library afterwards: it is 36 MB of push per gpurun call)
set -e
cd "$(dirname "$0")/../synference_amd/csrc"
make -j8 OBJDIR=/tmp/sf_fuzz_obj OUT=../lib/libsf_fuzz.so EXTRA=-DSF_FUZZ_SCHED 2>&1 | grep -E " error |Error |error:|libsf_fuzz" || true
ls -la ../lib/libsf_fuzz.so
