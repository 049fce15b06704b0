#!/bin/bash
# row split 262144 x 4096 (configs[4]) at world 1: rows per wide k_gemm_tn2 workgroup (the lane's single-workgroup kernels wait for a CU)
mkdir -p gpurun_out
for F in 0 16384 8192 4096 2048 0 4096; do DHQR_TN_WG_ROWS=$F timeout 300 python bench.py --config tallskinny --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tn_wg_rows', '$F', 'ms', d['ms_per_step'], 'GFLOP/s', d['value'], 'resid', d.get('residual'), d['phase_ms_per_step'])"; done | tee gpurun_out/r3g_tn_wg_rows.txt
