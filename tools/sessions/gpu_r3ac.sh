#!/bin/bash
# round 3, call ac: wide subtraction launches in chunks (DHQR_NN_SPLIT, default 4 with >= 48 tiles per chunk)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ac; mkdir -p $O; cd $R
for n in 1 4; do
  echo "== DHQR_NN_SPLIT=$n"
  DHQR_NN_SPLIT=$n timeout 600 python tools/quick_bench.py 32768,128 16384,128 8192,128 12288,128 128,128,262144 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' ', d['m'],'x',d['n'],'ms', round(d['t0']*1e3,2), round(d['t1']*1e3,2), 'GFLOP/s', round(d['gflops'],1), 'resid', d.get('resid'), 'ms_panel', d['stats'].get('ms_panel'))"
  DHQR_NN_SPLIT=$n timeout 600 python bench.py --config tallskinny --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  row split', d['config']['m'],'x',d['config']['n'],'ms', round(d['ms_per_step'],2), 'GFLOP/s', round(d['value'],1), 'resid', d['residual'])"
done > $O/ab_default.txt 2>&1
cat $O/ab_default.txt
