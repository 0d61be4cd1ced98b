#!/bin/bash
# The host side of the library under AddressSanitizer (the device code is not instrumented): an A/B build with -fsanitize=address, the CPU test suites that drive
# host-only entry points (shard reduces, TREC writer, index file checks, the whole sharded control flow over gloo with injected failures) with the runtime preloaded.
# alloc_dealloc_mismatch is off: the library's private operator new / delete (abi.cpp: malloc / free) pair with the sanitizer's interposed operators in this build only.
# usage: bash tools/asan_cpu.sh [ubsan]     (ubsan: -fsanitize=undefined instead; any "runtime error" line is a finding)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = ubsan ]; then
  bash tools/ab_build.sh ubsan "-fsanitize=undefined -fno-omit-frame-pointer -g -O1" 2>&1 | tail -1
  RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)
  export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0
  DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_ubsan.so LD_PRELOAD=$RT python -m pytest tests/test_cabi.py tests/test_trec_writer.py tests/test_abi_guard.py tests/test_host_logic.py tests/test_trec_helpers.py tests/test_dist_gloo.py -x -q -s 2>&1 | grep -E "passed|failed|runtime error" | sort | uniq -c
  exit 0
fi
bash tools/ab_build.sh asan "-fsanitize=address -fno-omit-frame-pointer -g -O1" 2>&1 | tail -1
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:allocator_may_return_null=1:alloc_dealloc_mismatch=0
DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_asan.so LD_PRELOAD=$RT python -m pytest tests/test_cabi.py tests/test_trec_writer.py tests/test_abi_guard.py tests/test_host_logic.py tests/test_trec_helpers.py tests/test_dist_gloo.py -x -q -s 2>&1 | grep -E "passed|failed|ERROR: AddressSanitizer|SUMMARY"
