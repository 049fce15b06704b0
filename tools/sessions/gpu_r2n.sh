#!/bin/bash
# 64-row tiles for the lane's narrow C -= V W products: parity + sizes with the switch off / on
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -3
for tr in 0 1 0 1; do
  echo "== DHQR_NN_TR64=$tr"
  DHQR_NN_TR64=$tr timeout 600 python tools/quick_bench.py 4096,128 8192,128 16384,128 2>&1 | grep "^{" | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['n'], 't1 %.2f ms' % (d['t1']*1e3), 'GFLOP/s %.0f' % d['gflops'], d['stats'].get('ms_panel'))
"
done 2>&1 | tee gpurun_out/n_tr64.txt
for tr in 0 1 0 1; do
  DHQR_NN_TR64=$tr timeout 600 python bench.py --no-cpu-baseline --no-residual --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tr64 $tr 32768: ms', round(d['ms_per_step'],2), d['phase_ms_per_step'])"
done 2>&1 | tee -a gpurun_out/n_tr64.txt
