#!/bin/bash
# kernel trace of the 8192^2 blocked factorisation (the regime where the panel chain, not the GEMMs, is the critical path)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_8192 -o q -- python $R/tools/quick_bench.py 8192,128 > $R/gpurun_out/prof_8192.log 2>&1
cd $R
python tools/prof_summary.py $(find gpurun_out/prof_8192 -name "*.db" | head -1) gpurun_out/q8192_kernel_stats.csv "python tools/quick_bench.py 8192,128 (2 factorisations)" | tail -1
python tools/prof_summary.py --by-stream $(find gpurun_out/prof_8192 -name "*.db" | head -1) gpurun_out/q8192_kernel_stats_by_stream.csv "python tools/quick_bench.py 8192,128 (2 factorisations)" | tail -1
find gpurun_out -name "*.db" -size +20M -delete
