#!/bin/bash
# AddressSanitizer run of the whole library on the CPU SIMT emulator (tests/simt): out-of-bounds reads that a GPU reports as a
# "memory access fault" without a location, and that the plain emulator tolerates (malloc slack), stop here with a stack.
# usage: tools/asan_emulated.sh <python script using dist_helpers.load_emulated_library("/tmp/asan/libdhqr_emulated.so")>
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
CL=/opt/rocm/lib/llvm/bin/clang++
mkdir -p /tmp/asan
$CL -x c++ -std=c++20 -O1 -g -DSIMT_FIBERS -fPIC -shared -fsanitize=address -shared-libasan -Wno-unknown-attributes -Wno-psabi \
    -Wno-unused-value -I $R/tests/simt/fake $R/distributedhouseholderqr.jl_amd/csrc/dhqr_api.hip \
    $R/distributedhouseholderqr.jl_amd/csrc/dhqr_unblocked.hip -o /tmp/asan/libdhqr_emulated.so
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 python "$@"
