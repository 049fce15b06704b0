#!/bin/bash
# NN kernel with the C tile streamed in during the K loop
mkdir -p gpurun_out
DHQR_NN_TIME=1 timeout 200 python tools/gemm_bench.py 0 32768 32768 3 0 16384 16384 5 0 8192 8192 5 2>&1 | grep -v amdgpu | tee gpurun_out/r2n_gemm.txt
timeout 300 python tools/quick_bench.py 32768,128 16384,128 8192,128 4400,128 2>&1 | grep -v amdgpu | tail -4 | cut -c1-700 | tee gpurun_out/r2n_full.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/r2n_pytest.txt
