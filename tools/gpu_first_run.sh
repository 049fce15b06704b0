#!/bin/bash
# first contact with the hardware: all GPU tests (no -x, so every failure is reported) + timing probe
mkdir -p gpurun_out
nproc > gpurun_out/nproc.txt; rocm-smi --showproductname 2>/dev/null | head -20 >> gpurun_out/nproc.txt
timeout 1700 python -m pytest tests -m gpu -q --timeout 600 -rfEs -p no:cacheprovider 2>&1 | tail -250 > gpurun_out/pytest_gpu.txt
tail -5 gpurun_out/pytest_gpu.txt
timeout 600 python tools/quick_bench.py > gpurun_out/quick_bench.txt 2>&1
cat gpurun_out/quick_bench.txt
