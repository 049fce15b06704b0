#!/bin/bash
# round 3, call x: kernel-boundary probe; A/B of the dynamic k_gemm_tn2 units and the 255-slot grid without the server
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3x; mkdir -p $O; cd $R
timeout 300 python tools/boundary_probe.py 2>&1 | grep -v amdgpu.ids > $O/boundary_probe.txt; cat $O/boundary_probe.txt
for cfg in "DHQR_SERVER=0 DHQR_LANE_FIT=0 DHQR_TN_DYNAMIC=0" "DHQR_SERVER=0 DHQR_LANE_FIT=0 DHQR_TN_DYNAMIC=1" "DHQR_SERVER=0 DHQR_LANE_FIT=0 DHQR_TN_DYNAMIC=0" "DHQR_SERVER=0 DHQR_LANE_FIT=0 DHQR_TN_DYNAMIC=1" "DHQR_SERVER=0 DHQR_LANE_FIT=1 DHQR_TN_DYNAMIC=0" "DHQR_SERVER=0 DHQR_LANE_FIT=0 DHQR_SPARE_CUS=1 DHQR_TN_DYNAMIC=1"; do
  echo "== $cfg"; env $cfg timeout 600 python tools/quick_bench.py 32768,128 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' ', d['m'],'x',d['n'],'ms', round(d['t0']*1e3,2), round(d['t1']*1e3,2), 'GFLOP/s', round(d['gflops'],1), 'ms_panel', d['stats'].get('ms_panel'))"
done > $O/ab.txt 2>&1
cat $O/ab.txt
