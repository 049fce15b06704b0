#!/bin/bash
# row split 262144 x 4096 at world 1 (bench.py --config tallskinny) with the narrow products in k_gemm_tn2 / k_gemm_tn workgroups
mkdir -p gpurun_out
{
for NT in 0 1 0 1; do
  DHQR_NARROW_TN=$NT python bench.py --config tallskinny --steps 3 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('narrow_tn $NT', d['ms_per_step'], d['value'])"
done
} > gpurun_out/r4z_rowsplit_narrow_tn.txt 2>&1
cat gpurun_out/r4z_rowsplit_narrow_tn.txt
