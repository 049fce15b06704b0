#!/bin/bash
# r4: narrow V'C through slot-sized k_gemm_tn workgroups (DHQR_NARROW_TN=1, the default) against k_gemm_tn2 (=0)
mkdir -p gpurun_out
{
for NT in 0 1 0 1; do
  DHQR_NARROW_TN=$NT python tools/lda_probe.py 32768 2>/dev/null | grep '^{' | head -1 | sed "s/^/narrow_tn $NT /"
done
for N in 8192 16384 24576; do for NT in 0 1; do
  DHQR_NARROW_TN=$NT python tools/lda_probe.py $N 2>/dev/null | grep '^{' | head -1 | sed "s/^/narrow_tn $NT /"
done; done
} > gpurun_out/r4t_narrow_tn.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r4t_pytest_parity.txt 2>&1
tail -3 gpurun_out/r4t_pytest_parity.txt
