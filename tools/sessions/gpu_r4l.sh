#!/bin/bash
# round 4, run L: randomised shapes on the final library (multi-rank drivers with quads at P > 1 forced on small matrices,
# the unblocked path), quads threshold at P > 1
mkdir -p gpurun_out
( timeout 900 python tools/gpu_fuzz.py 4 60 2>&1 | grep -v amdgpu | tail -25 ) > gpurun_out/r4l_fuzz_multirank.txt
( DHQR_QUAD_MIN_COLS=0 DHQR_TN_MODEL_MIN_TILES=2 timeout 900 python tools/gpu_fuzz.py 5 40 2>&1 | grep -v amdgpu | tail -12 ) > gpurun_out/r4l_fuzz_multirank_quads_streamk.txt
( timeout 600 python tools/gpu_fuzz_unblocked.py 7 40 2>&1 | grep -v amdgpu | tail -6 ) > gpurun_out/r4l_fuzz_unblocked.txt
for Q in 10240 4096; do for R in 2 4; do
  DHQR_QUAD_MIN_COLS=$Q python bench.py --logical-ranks $R --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('quad_min_cols', $Q, 'ranks', $R, 'ms', round(d['ms_per_step'],1))"
done; done > gpurun_out/r4l_quad_threshold.txt 2>&1
