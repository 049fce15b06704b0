#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -rfEs -p no:cacheprovider -x 2>&1 | tail -30 > gpurun_out/pytest_gpu3.txt
tail -4 gpurun_out/pytest_gpu3.txt
for P in 1 2; do echo "== DHQR_PANEL=$P"; DHQR_PANEL=$P timeout 600 python tools/quick_bench.py 8192,128 16384,128 32768,128 2>&1 | grep -v amdgpu.ids; done > gpurun_out/quick_bench3.txt
cat gpurun_out/quick_bench3.txt
for IB in 16 64; do echo "== DHQR_IB=$IB"; DHQR_IB=$IB timeout 600 python tools/quick_bench.py 16384,128 32768,128 2>&1 | grep -v amdgpu.ids; done > gpurun_out/quick_bench3b.txt
cat gpurun_out/quick_bench3b.txt
