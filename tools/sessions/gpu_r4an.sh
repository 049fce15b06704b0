#!/bin/bash
# r4: the panel commit on the wide stream (DHQR_COMMIT_WIDE=1) against on the lane's chain (0)
mkdir -p gpurun_out
{
for CW in 0 1 0 1; do
  DHQR_COMMIT_WIDE=$CW python tools/lda_probe.py 32768 2>/dev/null | grep '^{' | head -1 | sed "s/^/commit_wide $CW /"
done
for N in 8192 16384 24576; do for CW in 0 1; do
  DHQR_COMMIT_WIDE=$CW python tools/lda_probe.py $N 2>/dev/null | grep '^{' | head -1 | sed "s/^/commit_wide $CW /"
done; done
} > gpurun_out/r4an_commit_wide.txt 2>&1
cat gpurun_out/r4an_commit_wide.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r4an_pytest.txt 2>&1; tail -3 gpurun_out/r4an_pytest.txt
