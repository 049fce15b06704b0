#!/bin/bash
# round-end evidence: gpu tests, smoke, contract bench (both configs), rocprof summaries
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -rfEs -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/final_pytest.txt
tail -3 gpurun_out/final_pytest.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -4 | tee gpurun_out/final_smoke.txt
timeout 900 python bench.py > gpurun_out/final_bench_blocked.json 2> gpurun_out/final_bench_blocked.err; tail -c 2500 gpurun_out/final_bench_blocked.json
timeout 600 python bench.py --config unblocked --no-cpu-baseline > gpurun_out/final_bench_unblocked.json 2> gpurun_out/final_bench_unblocked.err; tail -c 1200 gpurun_out/final_bench_unblocked.json
timeout 300 python tools/c64_bench.py 8192 64 2>&1 | grep -v amdgpu | tail -1 | tee gpurun_out/final_c64_blocked8192.json
timeout 300 python tools/c64_bench.py 8192 0 2>&1 | grep -v amdgpu | tail -1 | tee gpurun_out/final_c64_unblocked8192.json
for R in 1 2 4 8; do timeout 300 python bench.py --logical-ranks $R --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('logical ranks', $R, 'ms', d['ms_per_step'], 'GFLOP/s', d['value'], 'resid', d.get('residual'))"; done | tee gpurun_out/final_logical_ranks.txt
timeout 300 python bench.py --config tallskinny --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final_bench_tallskinny.json 2> gpurun_out/final_bench_tallskinny.err; tail -c 900 gpurun_out/final_bench_tallskinny.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final_blocked -o blocked -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-residual > $R/gpurun_out/prof_final_blocked.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final_unblocked -o unblocked -- python $R/bench.py --config unblocked --steps 1 --warmup 1 --no-cpu-baseline --no-residual > $R/gpurun_out/prof_final_unblocked.log 2>&1
cd $R
python tools/prof_summary.py $(find gpurun_out/prof_final_blocked -name "*.db" | head -1) gpurun_out/final_blocked_kernel_stats.csv "python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-residual (2 factorisations in the trace)" | tail -1
python tools/prof_summary.py $(find gpurun_out/prof_final_unblocked -name "*.db" | head -1) gpurun_out/final_unblocked_kernel_stats.csv "python bench.py --config unblocked --steps 1 --warmup 1 --no-cpu-baseline --no-residual (2 factorisations in the trace)" | tail -1
find gpurun_out -name "*.db" -size +20M -delete
