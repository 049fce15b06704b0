#!/bin/bash
mkdir -p gpurun_out
( cd tools && ./top_probe 200 ) > gpurun_out/r4aa_top_probe.txt 2>&1
cat gpurun_out/r4aa_top_probe.txt
{
for LA in 0 1 0 1; do
  DHQR_PANEL_TOP_LA=$LA python tools/lda_probe.py 32768 2>/dev/null | grep '^{' | head -1 | sed "s/^/panel_top_la $LA /"
done
for N in 8192 16384; do for LA in 0 1; do
  DHQR_PANEL_TOP_LA=$LA python tools/lda_probe.py $N 2>/dev/null | grep '^{' | head -1 | sed "s/^/panel_top_la $LA /"
done; done
} > gpurun_out/r4aa_panel_top_la.txt 2>&1
cat gpurun_out/r4aa_panel_top_la.txt
