#!/bin/bash
# round 3, call ae: row chunks of the single-panel subtraction (blocked ComplexF64 is lane-bound: do drain points help its whole-CU panel workgroups?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ae; mkdir -p $O; cd $R
for n in 1 2 4 8; do
  echo "== DHQR_PA_CHUNKS=$n"
  ( DHQR_PA_CHUNKS=$n timeout 300 python tools/c64_bench.py 8192 64; DHQR_PA_CHUNKS=$n timeout 300 python tools/c64_bench.py 4096 64 ) 2>/dev/null | grep '^{' | cut -c1-120
  DHQR_PA_CHUNKS=$n timeout 300 python tools/quick_bench.py 8192,128 4096,128 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' ', d['m'],'x',d['n'],'ms', round(d['t1']*1e3,2), 'GFLOP/s', round(d['gflops'],1))"
done > $O/ab.txt 2>&1
cat $O/ab.txt
