#!/bin/bash
# unblocked path, tall-skinny shapes with m <= 8192: reflectors per pass against the lead workgroup's chain
mkdir -p gpurun_out
for K in 2 3 4 5; do echo "DHQR_RANKK=$K"; DHQR_RANKK=$K timeout 600 python tools/quick_bench.py 512,0,8192 1024,0,8192 2048,0,8192 4096,0,8192 1024,0,4096 2048,0,4096 3000,0,6000 128,0,8192 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' ', d['m'],'x',d['n'],'ms', round(d['t1']*1e3,2), 'GFLOP/s', round(d['gflops'],1))"; done | tee gpurun_out/r3k_unblocked_tall_K.txt
