#!/bin/bash
# round 3, call ab: per-launch trace of the blocked ComplexF64 factorisation (8192^2, nb = 64)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ab; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
( cd $R; timeout 600 rocprofv3 --kernel-trace -d $O/prof -o out -- python tools/c64_bench.py 8192 64 > $O/run.txt 2> $O/run.err )
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/prof_summary.py --per-launch $DB $O/c64_per_launch.csv "python tools/c64_bench.py 8192 64" | tail -1
python tools/prof_summary.py --by-stream $DB $O/c64_by_stream.csv "python tools/c64_bench.py 8192 64 (2 factorisations)" | tail -1
gzip -f $O/c64_per_launch.csv; find $O -name "*.db" -delete
cat $O/c64_by_stream.csv | cut -c1-160 | head -40; cat $O/run.txt | cut -c1-200
