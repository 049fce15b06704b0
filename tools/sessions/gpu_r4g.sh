#!/bin/bash
# round 4, run G: which of the round's changes moved the logical-rank runs (ranks sharing ONE GPU)?
mkdir -p gpurun_out
for R in 2 8; do for LS in 1 0; do for SK in 1 0; do
  DHQR_LANE_SIDE=$LS DHQR_TN_STREAMK=$SK python bench.py --logical-ranks $R --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('ranks', $R, 'lane_side', $LS, 'streamk', $SK, 'ms', round(d['ms_per_step'],1), 'GFLOP/s', round(d['value']))"
done; done; done > gpurun_out/r4g_logical_ab.txt 2>&1
