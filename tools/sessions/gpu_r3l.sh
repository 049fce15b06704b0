#!/bin/bash
# lead workgroup as a function of its own: parity, 8192^2, tall-skinny shapes
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "unblocked" -p no:cacheprovider 2>&1 | tail -3
for F in 5 5; do DHQR_RANKK=$F timeout 300 python bench.py --config unblocked --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rankk', '$F', 'ms', d['ms_per_step'], 'GFLOP/s', d['value'], 'resid', d.get('residual'))"; done
for K in 3 5; do echo "DHQR_RANKK=$K"; DHQR_RANKK=$K timeout 600 python tools/quick_bench.py 512,0,8192 2048,0,8192 4096,0,8192 2048,0,4096 3000,0,6000 4096,0 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' ', d['m'],'x',d['n'],'ms', round(d['t1']*1e3,2), 'GFLOP/s', round(d['gflops'],1))"; done
