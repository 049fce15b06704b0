#!/bin/bash
# DPP wave reductions: full GPU suite, then TSQR-HR / unblocked timings
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/r2q_pytest.txt
for cfg in "DHQR_TSQR=1"; do
  echo "== $cfg" | tee -a gpurun_out/r2q_timing.txt
  env $cfg timeout 300 python tools/quick_bench.py 8192,128 32768,128 2>&1 | grep -v amdgpu | tail -2 | cut -c1-300 | tee -a gpurun_out/r2q_timing.txt
  env $cfg timeout 300 python bench.py --config tallskinny --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tallskinny', d['ms_per_step'], 'ms', d['value'], 'GFLOP/s resid', d['residual'], d['phase_ms_per_step'], d['panels_fast_fallback'])" | tee -a gpurun_out/r2q_timing.txt
done
timeout 300 python bench.py --config unblocked --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400 | tee -a gpurun_out/r2q_timing.txt
timeout 300 python tools/quick_bench.py 8192,128 2>&1 | grep -v amdgpu | tail -1 | cut -c1-300 | tee -a gpurun_out/r2q_timing.txt
