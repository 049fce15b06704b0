#!/bin/bash
# ComplexF64 blocked driver with / without look-ahead + the complex GPU tests
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_complex.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/u_pytest_c64.txt
tail -3 gpurun_out/u_pytest_c64.txt
for la in 1 0; do for n in 8192 16384; do
  echo "== DHQR_LOOKAHEAD=$la n=$n"; DHQR_LOOKAHEAD=$la timeout 300 python tools/c64_bench.py $n 64 2>&1 | grep -v amdgpu | tail -1
done; done 2>&1 | tee gpurun_out/u_c64_bench.txt
