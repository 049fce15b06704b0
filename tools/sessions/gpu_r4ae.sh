#!/bin/bash
# r4 final tree: quad threshold sweep at 32768^2 and 16384^2 / 24576^2 (columns to the right of a quad below which pairs take over)
mkdir -p gpurun_out
{
for Q in 10240 6144 8192 12288 4096 10240; do
  DHQR_QUAD_MIN_COLS=$Q python tools/lda_probe.py 32768 2>/dev/null | grep '^{' | head -1 | sed "s/^/quad_min_cols $Q /"
done
for N in 16384 24576; do for Q in 10240 6144; do
  DHQR_QUAD_MIN_COLS=$Q python tools/lda_probe.py $N 2>/dev/null | grep '^{' | head -1 | sed "s/^/quad_min_cols $Q /"
done; done
} > gpurun_out/r4ae_quad_min_cols.txt 2>&1
cat gpurun_out/r4ae_quad_min_cols.txt
