#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -rfEs -p no:cacheprovider -x 2>&1 | tail -30 > gpurun_out/pytest_gpu4.txt
tail -4 gpurun_out/pytest_gpu4.txt
timeout 600 python tools/quick_bench.py 8192,128 16384,128 32768,128 2>&1 | grep -v amdgpu.ids > gpurun_out/quick_bench4.txt
cat gpurun_out/quick_bench4.txt
