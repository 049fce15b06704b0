#!/bin/bash
# Build libdhqr.so (the product: include/dhqr.h) and libdhqr_bench.so (the same sources + the micro-benchmarks of
# include/dhqr_bench.h, -DDHQR_BENCH_BUILD) for gfx950; cross-compiles without a GPU.  Usage: build.sh [extra hipcc flags]
# Translation units (dhqr_internal.h): dhqr_api.hip (twice: product / bench superset) and dhqr_unblocked.hip (once, shared by
# both libraries) compile side by side; every exit code is collected before the script reports.
# DHQR_BUILD_INCREMENTAL=1: units whose object is newer than every source of csrc/ and include/ are not recompiled
# (developer loop; the driver's build check always compiles everything).
set -uo pipefail
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result"
OBJ=../build_obj
mkdir -p $OBJ
newest=$(ls -t *.h *.hip ../../include/*.h | head -1)
deps_unblocked="dhqr_unblocked.hip dhqr_internal.h dhqr_rank1.h dhqr_common.h ../../include/dhqr.h"
fresh() {  # fresh <object> <sources...>: incremental mode and the object is newer than all of them
  [ "${DHQR_BUILD_INCREMENTAL:-0}" = "1" ] && [ $# -eq 1 ] && [ -f "$1" ] && [ "$1" -nt "$newest" ] && return 0
  if [ "${DHQR_BUILD_INCREMENTAL:-0}" = "1" ] && [ -f "$1" ]; then
    local o=$1; shift
    for s in "$@"; do [ "$o" -nt "$s" ] || return 1; done
    return 0
  fi
  return 1
}
pids=(); names=()
compile() {  # compile <object> <source> [flags...]
  local o=$1 s=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS "$@" -c "$s" -o "$o.tmp" && mv -f "$o.tmp" "$o" &
  pids+=($!); names+=("$o")
}
fresh $OBJ/api.o *.h dhqr_api.hip ../../include/dhqr.h || compile $OBJ/api.o dhqr_api.hip "$@"
fresh $OBJ/api_bench.o *.h dhqr_api.hip ../../include/dhqr.h ../../include/dhqr_bench.h || compile $OBJ/api_bench.o dhqr_api.hip -DDHQR_BENCH_BUILD "$@"
fresh $OBJ/unblocked.o $deps_unblocked || compile $OBJ/unblocked.o dhqr_unblocked.hip "$@"
rc=0
for i in "${!pids[@]}"; do
  wait "${pids[$i]}" || { echo "build.sh: hipcc failed for ${names[$i]}" >&2; rc=1; }
done
if [ $rc -ne 0 ]; then rm -f $OBJ/*.tmp; exit 1; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/api.o $OBJ/unblocked.o -o ../libdhqr.so.tmp &
P1=$!
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/api_bench.o $OBJ/unblocked.o -o ../libdhqr_bench.so.tmp &
P2=$!
wait $P1; R1=$?
wait $P2; R2=$?
if [ $R1 -ne 0 ] || [ $R2 -ne 0 ]; then
  rm -f ../libdhqr.so.tmp ../libdhqr_bench.so.tmp
  echo "build.sh: link failed (libdhqr.so: exit $R1, libdhqr_bench.so: exit $R2)" >&2
  exit 1
fi
mv -f ../libdhqr.so.tmp ../libdhqr.so
mv -f ../libdhqr_bench.so.tmp ../libdhqr_bench.so
echo "built $(readlink -f ../libdhqr.so) and $(readlink -f ../libdhqr_bench.so)"
