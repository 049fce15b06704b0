#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -rfEs -p no:cacheprovider -k "column_cyclic or microbench" 2>&1 | tail -40 > gpurun_out/pytest_gpu2.txt
tail -3 gpurun_out/pytest_gpu2.txt
python tools/issue_probe.py > gpurun_out/issue_probe.txt 2>&1; cat gpurun_out/issue_probe.txt
timeout 900 python bench.py > gpurun_out/bench_blocked.json 2> gpurun_out/bench_blocked.err; tail -c 3000 gpurun_out/bench_blocked.json; tail -5 gpurun_out/bench_blocked.err
timeout 600 python bench.py --config unblocked --no-cpu-baseline > gpurun_out/bench_unblocked.json 2> gpurun_out/bench_unblocked.err; tail -c 2000 gpurun_out/bench_unblocked.json; tail -5 gpurun_out/bench_unblocked.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_blocked -o blocked -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-residual > $R/gpurun_out/prof_blocked.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_unblocked -o unblocked -- python $R/bench.py --config unblocked --steps 1 --warmup 0 --no-cpu-baseline --no-residual > $R/gpurun_out/prof_unblocked.log 2>&1
cd $R; find gpurun_out -name "*stats*" | head; ls -la gpurun_out/prof_blocked/* | head -20
# keep only the small summaries (the raw traces can be tens of MB)
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
