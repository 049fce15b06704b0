#!/bin/bash
# round 3, call aa: k_rankk_fused with the lead as K pipelined workgroups (DHQR_RANKK_PIPE=1) against the one-workgroup lead
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3aa; mkdir -p $O; cd $R
for cfg in "DHQR_RANKK_PIPE=0" "DHQR_RANKK_PIPE=1" "DHQR_RANKK_PIPE=2"; do
  echo "== $cfg"
  env $cfg timeout 300 python tools/tall_check.py 8192,64 4100,70 1000,50 300,40 6001,33 2049,128 2>&1 | grep -v amdgpu.ids
  env $cfg timeout 600 python tools/quick_bench.py 8192,0 2048,0,8192 4096,0 2048,0 1024,0,4096 512,0,2048 6144,0 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' ', d['m'],'x',d['n'],'ms', round(d['t1']*1e3,2), 'GFLOP/s', round(d['gflops'],1), 'resid', d.get('resid'))"
done > $O/ab.txt 2>&1
cat $O/ab.txt
