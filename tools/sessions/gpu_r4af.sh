#!/bin/bash
# row split 262144 x 4096 at world 1: chunks of the wide subtraction launches (DHQR_NN_SPLIT: 1 = one launch, 4 = default) on the final tree
mkdir -p gpurun_out
{
for NS in 4 1 2 8 16 4; do
  DHQR_NN_SPLIT=$NS python bench.py --config tallskinny --steps 3 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r={e['kernel'][:14]: round(e['ms_per_step'],1) for e in d.get('roofline_all',[])}; print('nn_split $NS', round(d['ms_per_step'],2), round(d['value']), r)"
done
} > gpurun_out/r4af_rowsplit_chunks.txt 2>&1
cat gpurun_out/r4af_rowsplit_chunks.txt
