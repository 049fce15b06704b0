#!/bin/bash
# round 3, call y: row split 262144 x 4096 at world 1: bench line (with the all-reduce counters) and kernel stats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --config tallskinny --steps 3 --warmup 1 --no-cpu-baseline"
( cd $R; timeout 600 $CMD > $O/bench_ts.json 2> $O/bench_ts.err ); tail -1 $O/bench_ts.json | cut -c1-1500
( cd $R; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o out -- $CMD --no-residual > $O/bench_ts_traced.json 2> $O/bench_ts_traced.err )
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/prof_summary.py $DB $O/ts_kernel_stats.csv "$CMD --no-residual (4 factorisations in the trace)" | tail -1
python tools/prof_summary.py --per-launch $DB $O/ts_per_launch.csv "$CMD" | tail -1; gzip -f $O/ts_per_launch.csv
find $O -name "*.db" -delete
head -30 $O/ts_kernel_stats.csv
