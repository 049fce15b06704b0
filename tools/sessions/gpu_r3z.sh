#!/bin/bash
# round 3, call z: per-launch profile of the unblocked path on 16384 x 4096 (k_rankk_tall)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3z; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
( cd $R; timeout 600 rocprofv3 --kernel-trace -d $O/prof -o out -- python tools/quick_bench.py 4096,0,16384 > $O/run.txt 2> $O/run.err )
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rankk_profile.py $DB 4096 5 2 > $O/rankk_tall_16384x4096.txt 2>&1
find $O -name "*.db" -delete
cat $O/rankk_tall_16384x4096.txt
