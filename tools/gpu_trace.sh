#!/bin/bash
# per-launch kernel trace of one 32768^2 factorisation + the wide stream's gaps; usage: gpu_trace.sh <tag> [env assignments...]
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-residual --no-also"
env "$@" timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/trace_$TAG -o $TAG -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-residual --no-also > $R/gpurun_out/trace_$TAG.log 2>&1
cd $R
DB=$(find gpurun_out/trace_$TAG -name "*.db" | head -1)
python tools/prof_summary.py --per-launch $DB gpurun_out/trace_${TAG}_per_launch.csv "$CMD [$*]" | tail -1
gzip -f gpurun_out/trace_${TAG}_per_launch.csv
python tools/lane_gaps.py gpurun_out/trace_${TAG}_per_launch.csv.gz --around 10 > gpurun_out/trace_${TAG}_gaps.txt 2>&1
head -20 gpurun_out/trace_${TAG}_gaps.txt
rm -rf gpurun_out/trace_$TAG
