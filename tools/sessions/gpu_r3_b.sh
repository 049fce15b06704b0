#!/bin/bash
# round 3, call b: per-launch trace of one 32768^2 factorisation; ComplexF64 ratio diagnostic; the tightened / new GPU tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b; mkdir -p $O
cd $R
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_gpu.txt
( timeout 600 python tools/c64_ratio_diag.py 2>&1 | tail -20 ) > $O/c64_ratio_diag.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/prof -o trace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-residual > $O/prof.log 2>&1
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/prof_summary.py --per-launch $DB $O/per_launch.csv "python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-residual"
python tools/prof_summary.py --by-stream $DB $O/by_stream.csv "python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-residual"
python tools/prof_summary.py $DB $O/kernel_stats.csv "python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-residual"
find $O/prof -name "*.db" -delete
gzip -f $O/per_launch.csv
cat $O/pytest_gpu.txt | tail -5; cat $O/c64_ratio_diag.txt; head -12 $O/kernel_stats.csv | cut -c1-160
