#!/bin/bash
# Build libdhqr.so (the product: include/dhqr.h) and libdhqr_bench.so (the same source + the micro-benchmarks of
# include/dhqr_bench.h, -DDHQR_BENCH_BUILD) for gfx950; cross-compiles without a GPU.  Usage: build.sh [extra hipcc flags]
# Both compilations run side by side; both exit codes are collected before the script reports (a failure of the first
# must not leave the second running with its message lost).
set -uo pipefail
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS "$@" dhqr_api.hip -o ../libdhqr.so.tmp &
P1=$!
/opt/rocm/bin/hipcc $FLAGS -DDHQR_BENCH_BUILD "$@" dhqr_api.hip -o ../libdhqr_bench.so.tmp &
P2=$!
wait $P1; R1=$?
wait $P2; R2=$?
if [ $R1 -ne 0 ] || [ $R2 -ne 0 ]; then
  rm -f ../libdhqr.so.tmp ../libdhqr_bench.so.tmp
  echo "build.sh: hipcc failed (libdhqr.so: exit $R1, libdhqr_bench.so: exit $R2)" >&2
  exit 1
fi
mv -f ../libdhqr.so.tmp ../libdhqr.so
mv -f ../libdhqr_bench.so.tmp ../libdhqr_bench.so
echo "built $(readlink -f ../libdhqr.so) and $(readlink -f ../libdhqr_bench.so)"
