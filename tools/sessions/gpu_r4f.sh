#!/bin/bash
# round 4, run F: lane side stream (DHQR_LANE_SIDE) A/B on one box + parity subset
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blocked or logical_ranks or rejected or fast_panel or full_size or wide_tn or darray or drop_in or golden" 2>&1 | tail -6 > gpurun_out/r4f_tests.log
for LS in 1 0 1 0; do
  DHQR_LANE_SIDE=$LS python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lane_side', $LS, 'ms', round(d['ms_per_step'],2), 'GFLOP/s', round(d['value']), 'resid', d['residual'], [(r['kernel'][:14], round(r['frac'],4), round(r['ms_per_step'],1)) for r in d['roofline_all']])"
done > gpurun_out/r4f_lane_side.txt 2>&1
for LS in 1 0; do DHQR_LANE_SIDE=$LS python tools/quick_bench.py 8192,128 16384,128 24576,128 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('lane_side', $LS, d['m'],'x',d['n'],'ms', round(d['t1']*1e3,2), 'GFLOP/s', round(d['gflops'],1), 'resid', d.get('resid'))"; done >> gpurun_out/r4f_lane_side.txt 2>&1
for LS in 1 0; do DHQR_LANE_SIDE=$LS python bench.py --config tallskinny --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lane_side', $LS, 'tallskinny ms', round(d['ms_per_step'],2), 'GFLOP/s', round(d['value']))"; done >> gpurun_out/r4f_lane_side.txt 2>&1
DHQR_HOSTIO_TRACE=1 python tools/hostio_bench.py 32768 2 > gpurun_out/r4f_hostio.log 2>&1
