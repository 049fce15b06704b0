#!/bin/bash
# round 4, run J: the quad head folded into the wide launches (default) against the separate head (DHQR_QUAD_HEAD=1)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blocked or full_size or rejected or fast_panel or wide_tn or drop_in" 2>&1 | tail -4 > gpurun_out/r4j_tests.log
for QH in 0 1 0 1; do
  DHQR_QUAD_HEAD=$QH python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('quad_head', $QH, 'ms', round(d['ms_per_step'],2), 'GFLOP/s', round(d['value']), 'resid', d['residual'], [(r['kernel'][:14], round(r['frac'],4), round(r['ms_per_step'],1)) for r in d['roofline_all']])"
done > gpurun_out/r4j_quad_head.txt 2>&1
for QH in 0 1; do DHQR_QUAD_HEAD=$QH python tools/quick_bench.py 16384,128 24576,128 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('quad_head', $QH, d['m'],'x',d['n'],'ms', round(d['t1']*1e3,2), 'GFLOP/s', round(d['gflops'],1), 'resid', d.get('resid'))"; done >> gpurun_out/r4j_quad_head.txt 2>&1
for R in 2 8; do python bench.py --logical-ranks $R --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('ranks', $R, 'ms', round(d['ms_per_step'],1), 'GFLOP/s', round(d['value']))"; done >> gpurun_out/r4j_quad_head.txt 2>&1
