#!/bin/bash
# A/B: rungs of the workgroup-size ladder of k_rankk_fused (bit 0: 896, 1: 768, 2: 640, 3: 384 switched off)
mkdir -p gpurun_out
for F in 0 1 2 4 8 5 15 0; do DHQR_RANKK_OFF=$F timeout 300 python bench.py --config unblocked --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rungs off mask', '$F', 'ms', d['ms_per_step'], 'GFLOP/s', d['value'], 'resid', d.get('residual'))"; done | tee gpurun_out/r3i_ladder_ab.txt
