#!/bin/bash
# round 3, call w: per-launch trace of 32768^2 with the panel server + small-footprint lane kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3w; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-residual"
( cd $R; timeout 600 rocprofv3 --kernel-trace -d $O/prof -o out -- $CMD > $O/bench_traced.json 2> $O/bench_traced.err )
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/prof_summary.py --per-launch $DB $O/per_launch.csv "$CMD" | tail -1
python tools/lane_gaps.py $O/per_launch.csv --around 25 > $O/lane_gaps.txt 2>&1
gzip -f $O/per_launch.csv; find $O -name "*.db" -delete
head -40 $O/lane_gaps.txt; tail -3 $O/bench_traced.json | cut -c1-300
