#!/bin/bash
mkdir -p gpurun_out
for LS in 0 1 0 1; do for R in 2 8; do
  DHQR_LANE_SIDE=$LS python bench.py --logical-ranks $R --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('lane_side(+stream)', $LS, 'ranks', $R, 'ms', round(d['ms_per_step'],1), 'GFLOP/s', round(d['value']))"
done; done > gpurun_out/r4i_logical.txt 2>&1
git -C . log --oneline | head -1 >> gpurun_out/r4i_logical.txt
