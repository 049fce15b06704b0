#!/bin/bash
# round 3, call ak: quad threshold re-check on the final code (same box)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ak; mkdir -p $O; cd $R
for q in 10240 6144 8192 12288 10240; do
  echo "== DHQR_QUAD_MIN_COLS=$q"
  DHQR_QUAD_MIN_COLS=$q timeout 600 python tools/quick_bench.py 32768,128 24576,128 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' ', d['m'],'x',d['n'],'ms', round(d['t0']*1e3,2), round(d['t1']*1e3,2), 'GFLOP/s', round(d['gflops'],1))"
done > $O/ab.txt 2>&1
cat $O/ab.txt
