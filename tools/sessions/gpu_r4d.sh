#!/bin/bash
# round 4, run D: stream-K k_gemm_tn2 A/B on one box; host IO phase trace
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "wide_tn or blocked_vs_oracle or full_size_properties" 2>&1 | tail -6 > gpurun_out/r4d_tests.log
for SK in 1 0 1 0; do
  DHQR_TN_STREAMK=$SK python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('streamk', $SK, 'ms', round(d['ms_per_step'],2), 'GFLOP/s', round(d['value']), 'resid', d['residual'], [(r['kernel'][:14], round(r['frac'],4), round(r['ms_per_step'],1)) for r in d['roofline_all']])"
done > gpurun_out/r4d_streamk.txt 2>&1
for SK in 1 0; do DHQR_TN_STREAMK=$SK python tools/quick_bench.py 16384,128 24576,128 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('streamk', $SK, d['m'],'x',d['n'],'ms', round(d['t1']*1e3,2), 'GFLOP/s', round(d['gflops'],1))"; done >> gpurun_out/r4d_streamk.txt 2>&1
DHQR_HOSTIO_TRACE=1 python tools/hostio_bench.py 32768 2 > gpurun_out/r4d_hostio.log 2>&1
