#!/bin/bash
# Build libdhqr.so (the product: include/dhqr.h) and libdhqr_bench.so (the same source + the micro-benchmarks of
# include/dhqr_bench.h, -DDHQR_BENCH_BUILD) for gfx950; cross-compiles without a GPU.  Usage: build.sh [extra hipcc flags]
set -euo pipefail
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS "$@" dhqr_api.hip -o ../libdhqr.so &
P1=$!
/opt/rocm/bin/hipcc $FLAGS -DDHQR_BENCH_BUILD "$@" dhqr_api.hip -o ../libdhqr_bench.so &
P2=$!
wait $P1
wait $P2
echo "built $(readlink -f ../libdhqr.so) and $(readlink -f ../libdhqr_bench.so)"
