#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3; do
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('prof_switch build: ms', round(d['ms_per_step'],2), 'GFLOP/s', round(d['value']), [(r['kernel'][:14], round(r['frac'],4), round(r['ms_per_step'],1), r['launches_per_step']) for r in d['roofline_all']], d['phase_ms_per_step'])"
done > gpurun_out/r4n_prof_switch.txt 2>&1
python -m pytest tests -m gpu -x -q -k "blocked_vs_oracle or fast_panel or logical_ranks or row_split or tall_skinny" 2>&1 | tail -3 >> gpurun_out/r4n_prof_switch.txt
