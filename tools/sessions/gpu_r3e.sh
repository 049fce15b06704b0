#!/bin/bash
# A/B: non-temporal column stores in the bulk of k_rankk_fused (do dirty L2 lines lengthen the gap between dependent launches?)
mkdir -p gpurun_out
for F in 0 1 2 3 1 3; do DHQR_RANKK_NT=$F timeout 300 python bench.py --config unblocked --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nt', '$F', 'ms', d['ms_per_step'], 'GFLOP/s', d['value'], 'resid', d.get('residual'), 'GB/s', d['roofline']['achieved'], 'event-timed kernel ms per step', d['roofline_all'][0]['total_ms'] / 5)"; done | tee gpurun_out/r3e_nt_ab.txt
