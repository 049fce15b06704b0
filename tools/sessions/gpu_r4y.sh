#!/bin/bash
mkdir -p gpurun_out
{
for HE in 0 1 0 1; do
  DHQR_HEAD_EARLY=$HE python tools/lda_probe.py 32768 2>/dev/null | grep '^{' | head -1 | sed "s/^/head_early $HE /"
done
for N in 16384 24576; do for HE in 0 1; do
  DHQR_HEAD_EARLY=$HE python tools/lda_probe.py $N 2>/dev/null | grep '^{' | head -1 | sed "s/^/head_early $HE /"
done; done
} > gpurun_out/r4y_head_early.txt 2>&1
cat gpurun_out/r4y_head_early.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blocked or unblocked" > gpurun_out/r4y_pytest.txt 2>&1; tail -3 gpurun_out/r4y_pytest.txt
