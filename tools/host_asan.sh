#!/bin/bash
# The library's host side (scene compiler, option and argument validation, program walk, launcher-free entry points)
# under AddressSanitizer + UBSan: builds csrc/libprt_hip_asan.so and runs the CPU test suite on it (no GPU: the
# sanitizers are host-only -- GPU ASan is not available on this pool).  The register-allocation guard is left
# out: this is an -O1 build.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C $ROOT/pyrayt_amd/csrc libprt_hip_asan.so
RT=$(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.asan-x86_64.so" | head -1)
cd $ROOT
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 LD_PRELOAD=$RT PRT_LIB=$ROOT/pyrayt_amd/csrc/libprt_hip_asan.so \
  python -m pytest tests -q -m "not gpu" -p no:cacheprovider \
  --deselect tests/test_abi.py::test_generation_kernel_keeps_its_register_allocation
