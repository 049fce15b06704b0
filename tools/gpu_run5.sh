#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -rfEs -p no:cacheprovider -x 2>&1 | tail -30 > gpurun_out/pytest_gpu5.txt
tail -4 gpurun_out/pytest_gpu5.txt
for LA in 0 1; do echo "== DHQR_LOOKAHEAD=$LA"; DHQR_LOOKAHEAD=$LA timeout 600 python tools/quick_bench.py 8192,128 16384,128 32768,128 2>&1 | grep -v amdgpu.ids; done > gpurun_out/quick_bench5.txt
cat gpurun_out/quick_bench5.txt
