#!/bin/bash
# rocprofv3 kernel trace of the device-resident solve at one shape; usage: gpu_r5_solve_prof.sh <m,n> <tag>
SHAPE=$1; TAG=$2
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o $TAG -- python $R/tools/solve_bench.py $SHAPE > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
python tools/prof_summary.py $(find gpurun_out/prof_$TAG -name "*.db" | head -1) gpurun_out/prof_${TAG}_kernel_stats.csv "python tools/solve_bench.py $SHAPE"
grep -v "at::\|elementwise\|Cijk\|gemv\|reduce_kernel" gpurun_out/prof_${TAG}_kernel_stats.csv | head -24 | cut -c1-160
grep "^{" gpurun_out/prof_$TAG.log | cut -c1-300
find gpurun_out/prof_$TAG -name "*.db" -size +30M -delete
