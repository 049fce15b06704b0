#!/bin/bash
# round 3, call v: panel server + small-footprint lane kernels: 32768^2 A/B, then the per-launch trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3v; mkdir -p $O; cd $R
for cfg in "DHQR_SERVER=0 DHQR_LANE_FIT=1" "DHQR_SERVER=1 DHQR_LANE_FIT=1"; do
  echo "== $cfg"; env $cfg timeout 600 python tools/quick_bench.py 32768,128 16384,128 8192,128 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' ', d['m'],'x',d['n'],'ms', round(d['t0']*1e3,2), round(d['t1']*1e3,2), 'GFLOP/s', round(d['gflops'],1), 'resid', d.get('resid'), 'ms_panel', d['stats'].get('ms_panel'))"
done > $O/ab.txt 2>&1
cat $O/ab.txt
cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-residual"
( cd $R; timeout 600 rocprofv3 --kernel-trace -d $O/prof -o out -- $CMD > $O/bench_traced.json 2> $O/bench_traced.err )
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/prof_summary.py --per-launch $DB $O/per_launch.csv "$CMD" | tail -1
gzip -f $O/per_launch.csv; find $O -name "*.db" -delete
tail -1 $O/bench_traced.json | cut -c1-400
# the pipelines whose polls became relaxed loads: ComplexF64 8192^2 blocked, tall unblocked
cd $R
(timeout 300 python tools/c64_bench.py 8192 64; timeout 300 python tools/c64_bench.py 4096 64) 2>/dev/null | grep "^{" | cut -c1-200 > $O/c64.txt; cat $O/c64.txt
timeout 300 python tools/quick_bench.py 4096,0,16384 2048,0,12288 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' ', d['m'],'x',d['n'],'ms', round(d['t1']*1e3,2), 'GFLOP/s', round(d['gflops'],1), 'resid', d.get('resid'))" > $O/tall.txt; cat $O/tall.txt
