#!/bin/bash
# cyclic block = a pair of panels: multi-rank GPU tests + logical ranks on the one GPU
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 -k "rank or multi or cyclic or darray or logical or rccl or column" -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/z_pytest.txt
for R in 1 2 4 8; do timeout 300 python bench.py --logical-ranks $R --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('logical ranks', $R, 'ms', d['ms_per_step'], 'GFLOP/s', d['value'], 'resid', d.get('residual'), d.get('per_rank_panel_ms_per_step'))"; done | tee gpurun_out/z_logical_ranks.txt
