#!/bin/bash
# round 3, call ad: wide k_gemm_tn2 launches in column chunks (DHQR_TN_SPLIT)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ad; mkdir -p $O; cd $R
for n in 1 2 3; do
  echo "== DHQR_TN_SPLIT=$n"
  DHQR_TN_SPLIT=$n timeout 600 python tools/quick_bench.py 32768,128 16384,128 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' ', d['m'],'x',d['n'],'ms', round(d['t0']*1e3,2), round(d['t1']*1e3,2), 'GFLOP/s', round(d['gflops'],1), 'resid', d.get('resid'), 'ms_panel', d['stats'].get('ms_panel'))"
done > $O/ab.txt 2>&1
cat $O/ab.txt
