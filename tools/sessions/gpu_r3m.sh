#!/bin/bash
# round 3, call m: k_rankk_tall (columns of 8192 < rows <= 16384, K reflectors per pass): parity vs oracle, sizes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m; mkdir -p $O; cd $R
for K in 1 3 5; do echo "== DHQR_RANKK_TALL=$K"; DHQR_RANKK_TALL=$K timeout 600 python tools/tall_check.py 2>&1 | grep -v amdgpu.ids; done > $O/tall_parity.txt 2>&1
for K in 1 3 5; do echo "== DHQR_RANKK_TALL=$K"; DHQR_RANKK_TALL=$K timeout 900 python tools/quick_bench.py 12288,0 4096,0,16384 2048,0,12288 16384,0 8192,0 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' ', d['m'],'x',d['n'],'ms', round(d['t1']*1e3,2), 'GFLOP/s', round(d['gflops'],1), 'resid', d.get('resid'))"; done > $O/tall_sizes.txt 2>&1
cat $O/tall_parity.txt $O/tall_sizes.txt
