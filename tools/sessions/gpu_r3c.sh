#!/bin/bash
# unblocked path replayed as a hipGraph (DHQR_GRAPH=1) against plain launches
mkdir -p gpurun_out
DHQR_RANKK=3 DHQR_GRAPH=1 timeout 300 python bench.py --config unblocked --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -5 | cut -c1-600
for F in "3 0" "3 1" "2 1" "1 1" "3 0" "3 1"; do set -- $F; DHQR_RANKK=$1 DHQR_GRAPH=$2 timeout 300 python bench.py --config unblocked --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rankk graph', '$F', 'ms', d['ms_per_step'], 'GFLOP/s', d['value'], 'resid', d.get('residual'), 'GB/s', d['roofline']['achieved'], d['roofline']['launches'])"; done | tee gpurun_out/r3c_graph_ab.txt
