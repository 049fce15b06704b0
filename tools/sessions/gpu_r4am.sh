#!/bin/bash
# r4 final tree: stream-K / split model of the wide k_gemm_tn2 launches from fewer column tiles (DHQR_TN_MODEL_MIN_TILES; default 128)
mkdir -p gpurun_out
{
for MT in 128 64 32 16 128 32; do
  DHQR_TN_MODEL_MIN_TILES=$MT python tools/lda_probe.py 32768 2>/dev/null | grep '^{' | head -1 | sed "s/^/tn_model_min_tiles $MT /"
done
for N in 16384 8192 24576; do for MT in 128 32 16; do
  DHQR_TN_MODEL_MIN_TILES=$MT python tools/lda_probe.py $N 2>/dev/null | grep '^{' | head -1 | sed "s/^/tn_model_min_tiles $MT /"
done; done
} > gpurun_out/r4am_tn_min_tiles.txt 2>&1
cat gpurun_out/r4am_tn_min_tiles.txt
