#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -rfEs -p no:cacheprovider -x -k "column_cyclic or golden" 2>&1 | tail -8 > gpurun_out/pytest_gpu7.txt
tail -4 gpurun_out/pytest_gpu7.txt
for D in python c; do timeout 900 python bench.py --driver $D --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$D', 'ms/step', round(d['ms_per_step'],1), 'TF/s', round(d['value']/1e3,2), 'resid', d['residual'], {k: round(v,1) for k,v in d['phase_ms_per_step'].items()})"; done | tee gpurun_out/bench7.txt
