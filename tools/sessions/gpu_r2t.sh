#!/bin/bash
# row-split driver A/B on the GPU box: pairs / look-ahead on and off, plus the row-split GPU tests
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "row_split or rowsplit or tall or rs_" -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/t_pytest_rs.txt
tail -3 gpurun_out/t_pytest_rs.txt
for cfg in "DHQR_PAIR=1 DHQR_LOOKAHEAD=1" "DHQR_PAIR=1 DHQR_LOOKAHEAD=0" "DHQR_PAIR=0 DHQR_LOOKAHEAD=1" "DHQR_PAIR=0 DHQR_LOOKAHEAD=0"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --config tallskinny --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms', round(d['ms_per_step'],2), 'GFLOP/s', round(d['value']), 'resid', d.get('residual'), d.get('phase_ms_per_step'), d.get('panels_fast_fallback'))"
done 2>&1 | tee gpurun_out/t_rs_ab.txt
for R in 2 4 8; do timeout 300 python bench.py --config tallskinny --logical-ranks $R --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('logical ranks', $R, 'ms', d['ms_per_step'], 'GFLOP/s', d['value'], 'resid', d.get('residual'))"; done 2>&1 | tee gpurun_out/t_rs_logical.txt
