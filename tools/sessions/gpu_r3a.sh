#!/bin/bash
# K reflectors per pass (k_rankk_fused): parity + timing at 8192^2 unblocked (BASELINE configs[1])
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "unblocked" -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/r3a_pytest.txt
for F in 5 5; do DHQR_RANKK=$F timeout 300 python bench.py --config unblocked --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rankk', '$F', 'ms', d['ms_per_step'], 'GFLOP/s', d['value'], 'resid', d.get('residual'), 'GB/s', d['roofline']['achieved'], d['roofline']['launches'])"; done | tee gpurun_out/r3a_ab.txt
