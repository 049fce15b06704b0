#!/bin/bash
# kernel trace (by stream) of the 262144 x 4096 row-split factorisation at world 1 (BASELINE configs[4]): what the panel lane spends
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_ts -o q -- python $R/bench.py --config tallskinny --steps 1 --warmup 1 --no-cpu-baseline --no-residual > $R/gpurun_out/prof_ts.log 2>&1
cd $R
python tools/prof_summary.py --by-stream $(find gpurun_out/prof_ts -name "*.db" | head -1) gpurun_out/r3f_tallskinny_kernel_stats_by_stream.csv "python bench.py --config tallskinny --steps 1 --warmup 1 --no-cpu-baseline --no-residual (2 factorisations)" | tail -1
python tools/prof_summary.py $(find gpurun_out/prof_ts -name "*.db" | head -1) gpurun_out/r3f_tallskinny_kernel_stats.csv "python bench.py --config tallskinny --steps 1 --warmup 1 --no-cpu-baseline --no-residual (2 factorisations)" | tail -1
find gpurun_out -name "*.db" -delete
