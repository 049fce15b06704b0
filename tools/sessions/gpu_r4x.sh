#!/bin/bash
mkdir -p gpurun_out
{
for SC in 0 1 2 0 1; do
  DHQR_SPARE_CUS=$SC python tools/lda_probe.py 32768 2>/dev/null | grep '^{' | head -1 | sed "s/^/spare_cus $SC /"
done
} > gpurun_out/r4x_spare_cus.txt 2>&1
cat gpurun_out/r4x_spare_cus.txt
