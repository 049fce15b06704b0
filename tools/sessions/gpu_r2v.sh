#!/bin/bash
# does overlapping the panel lane with the wide update pay at small trailing sizes?  (tail of the 32768^2 run)
mkdir -p gpurun_out
for la in 1 0; do
  echo "== DHQR_LOOKAHEAD=$la"
  DHQR_LOOKAHEAD=$la timeout 600 python tools/quick_bench.py 4096,128 8192,128 12288,128 16384,128 24576,128 2>&1 | grep -v amdgpu | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    d = json.loads(ln); print(d['n'], 't0 %.2f ms t1 %.2f ms' % (d['t0']*1e3, d['t1']*1e3), 'GFLOP/s %.0f' % d['gflops'], d['stats'].get('ms_panel'), d['stats'].get('ms_gemm_vta'), d['stats'].get('ms_gemm_avw'))
"
done 2>&1 | tee gpurun_out/v_la_sizes.txt
