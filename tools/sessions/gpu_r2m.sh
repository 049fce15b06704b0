#!/bin/bash
# wide k_gemm_tn2 launches: split-K factor from the round / partial-traffic estimate (for >= N column tiles) vs round filling only
mkdir -p gpurun_out
for mt in 100000 128 100000 128 100000 128 3; do
  DHQR_TN_MODEL_MIN_TILES=$mt timeout 600 python bench.py --no-cpu-baseline --no-residual --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tn model for ntiles >= $mt 32768: ms', round(d['ms_per_step'],2), 'GFLOP/s', round(d['value']), d['phase_ms_per_step'])"
done 2>&1 | tee gpurun_out/m_tn_model3.txt
