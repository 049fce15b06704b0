#!/bin/bash
# Build libdhqr.so for gfx950 (cross-compiles without a GPU). Usage: build.sh [extra hipcc flags]
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libdhqr.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
  -Wall -Wno-unused-function -Wno-unused-result "$@" dhqr_api.hip -o "$OUT"
echo "built $(readlink -f $OUT)"
