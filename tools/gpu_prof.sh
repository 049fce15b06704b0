#!/bin/bash
# rocprofv3 kernel trace of one bench step; usage: gpu_prof.sh <tag> [env assignments...] 
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o $TAG -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-residual > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
python tools/prof_summary.py $(find gpurun_out/prof_$TAG -name "*.db" | head -1) gpurun_out/prof_${TAG}_kernel_stats.csv "python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-residual  [$*]"
head -30 gpurun_out/prof_${TAG}_kernel_stats.csv | cut -c1-150
tail -2 gpurun_out/prof_$TAG.log | cut -c1-600
find gpurun_out/prof_$TAG -name "*.db" -size +30M -delete
