#!/bin/bash
# round 4, run H: hardware queues (GPU_MAX_HW_QUEUES) against the number of streams the drivers use
mkdir -p gpurun_out
for Q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$Q python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('hwq', $Q, 'ranks 1 ms', round(d['ms_per_step'],1), 'GFLOP/s', round(d['value']))"
  for R in 2 8; do
  GPU_MAX_HW_QUEUES=$Q python bench.py --logical-ranks $R --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('hwq', $Q, 'ranks', $R, 'ms', round(d['ms_per_step'],1), 'GFLOP/s', round(d['value']))"
  done
done > gpurun_out/r4h_hwq.txt 2>&1
