#!/bin/bash
# round 3, call af: per-launch trace of 16384^2 blocked (where does a mid-size factorisation wait?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3af; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
( cd $R; timeout 600 rocprofv3 --kernel-trace -d $O/prof -o out -- python tools/quick_bench.py 16384,128 > $O/run.txt 2> $O/run.err )
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/prof_summary.py --per-launch $DB $O/per_launch_16384.csv "python tools/quick_bench.py 16384,128" | tail -1
gzip -f $O/per_launch_16384.csv; find $O -name "*.db" -delete
grep '^{' $O/run.txt | cut -c1-400
