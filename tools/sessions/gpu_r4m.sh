#!/bin/bash
mkdir -p gpurun_out
for PL in 0 1 0 1; do
  DHQR_PROFILE_LANE=$PL python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('profile_lane', $PL, 'ms', round(d['ms_per_step'],2), 'GFLOP/s', round(d['value']), [(r['kernel'][:14], round(r['frac'],4), round(r['ms_per_step'],1), r['launches_per_step']) for r in d['roofline_all']])"
done > gpurun_out/r4m_profile_lane.txt 2>&1
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blocked_vs_oracle or fast_panel or logical_ranks" 2>&1 | tail -3 >> gpurun_out/r4m_profile_lane.txt
