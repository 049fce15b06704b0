#!/bin/bash
mkdir -p gpurun_out
for OS in 1 0 1 0; do
  DHQR_OWN_STREAM=$OS python tools/lda_probe.py 32768 2>/dev/null | grep '^{' | head -1 | sed "s/^/own_stream $OS /"
done > gpurun_out/r4q_own_stream.txt 2>&1
