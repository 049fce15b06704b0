#!/bin/bash
# round 3, call ah: more than five reflectors per pass below 8193 rows through the streaming kernel (DHQR_RANKK_STREAM = 6..8)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ah; mkdir -p $O; cd $R
for cfg in "DHQR_RANKK_STREAM=0" "DHQR_RANKK_STREAM=6" "DHQR_RANKK_STREAM=8"; do
  echo "== $cfg"
  env $cfg timeout 300 python tools/tall_check.py 8192,64 4100,70 1000,50 6001,33 2049,128 2>&1 | grep -v amdgpu.ids
  env $cfg timeout 600 python tools/quick_bench.py 8192,0 2048,0,8192 4096,0 6144,0 12288,0 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' ', d['m'],'x',d['n'],'ms', round(d['t1']*1e3,2), 'GFLOP/s', round(d['gflops'],1), 'resid', d.get('resid'))"
done > $O/ab.txt 2>&1
cat $O/ab.txt
