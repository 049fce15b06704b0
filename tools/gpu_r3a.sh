#!/bin/bash
# K reflectors per pass (k_rankk_fused): parity + A/B at 8192^2 unblocked (BASELINE configs[1])
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -m gpu -q -x --timeout 600 -k "unblocked or zero_pivot or golden or reference" -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/r3a_pytest.txt
for F in 1 2 3 4 1 2 3 4; do DHQR_RANKK=$F timeout 300 python bench.py --config unblocked --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rankk', $F, 'ms', d['ms_per_step'], 'GFLOP/s', d['value'], 'resid', d.get('residual'), d['roofline'])"; done | tee gpurun_out/r3a_ab.txt
