#!/bin/bash
mkdir -p gpurun_out
for n in 2048 4096 6144 8192 12288; do for pm in 0 1000000; do
  DHQR_PAIR_MIN_N=$pm timeout 300 python tools/quick_bench.py $n,128 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n', d['n'], 'pair_min_n', $pm, 't1 ms', round(d['t1']*1e3,2), 'resid', d.get('resid'))"
done; done 2>&1 | tee gpurun_out/y_pair_min.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -3
