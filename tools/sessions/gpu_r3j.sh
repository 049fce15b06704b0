#!/bin/bash
# unblocked path over sizes: one reflector per launch (DHQR_RANKK=1) against five per pass (default)
mkdir -p gpurun_out
for K in 1 5; do echo "DHQR_RANKK=$K"; DHQR_RANKK=$K timeout 600 python tools/quick_bench.py 1024,0 2048,0 4096,0 8192,0 2048,0,8192 4096,0,16384 12288,0 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' ', d['m'],'x',d['n'],'ms', round(d['t1']*1e3,2), 'GFLOP/s', round(d['gflops'],1), 'resid', d.get('resid'))"; done | tee gpurun_out/r3j_unblocked_sizes.txt
