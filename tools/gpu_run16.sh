#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x -k "golden or column_cyclic_driver_single" 2>&1 | tail -3
timeout 900 python tools/quick_bench.py 65536,128 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/quick_bench_65536.txt | cut -c1-900
