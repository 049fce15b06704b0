#!/bin/bash
# round 3 evidence on the GPU box: gpu tests, smoke, the contract bench line + the other configurations, rocprofv3
# kernel-trace summaries of the same commands, HBM counter passes (one counter per pass) -> gpurun_out/r3e/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3e4; mkdir -p $O; cd $R
( timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -rfEs -p no:cacheprovider 2>&1 | tail -15 ) > $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
( timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -4 ) > $O/smoke.txt; cat $O/smoke.txt
timeout 900 python bench.py > $O/bench_blocked32768.json 2> $O/bench_blocked32768.err; tail -c 600 $O/bench_blocked32768.json
timeout 600 python bench.py --config unblocked --no-cpu-baseline > $O/bench_unblocked8192.json 2> $O/bench_unblocked8192.err
timeout 300 python bench.py --config tallskinny --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_tallskinny262144x4096.json 2> $O/bench_tallskinny.err
( for nb in 64 0; do timeout 300 python tools/c64_bench.py 8192 $nb 2>&1 | grep -v amdgpu | tail -1; done ) > $O/bench_complex8192.txt
( for LR in 1 2 4 8; do timeout 300 python bench.py --logical-ranks $LR --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('logical ranks', $LR, 'ms', d['ms_per_step'], 'GFLOP/s', d['value'], 'resid', d.get('residual'))"; done ) > $O/logical_ranks.txt
( timeout 600 python tools/quick_bench.py 12288,0 4096,0,16384 2048,0,8192 16384,0 8192,0 8192,128 16384,128 24576,128 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['m'],'x',d['n'],'nb',d['nb'],'ms', round(d['t1']*1e3,2), 'GFLOP/s', round(d['gflops'],1), 'resid', d.get('resid'))" ) > $O/sizes.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_blocked -o blocked -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-residual > $O/prof_blocked.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_unblocked -o unblocked -- python $R/bench.py --config unblocked --steps 1 --warmup 1 --no-cpu-baseline --no-residual > $O/prof_unblocked.log 2>&1
D=$R/tools/pmc_driver
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( timeout 500 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc/blocked_$ctr -o out --output-format csv -- $D blocked 32768 > $O/pmc_blocked_$ctr.log 2>&1; echo "rc=$?" >> $O/pmc_blocked_$ctr.log )
  tail -1 $O/pmc_blocked_$ctr.log
  ( timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc/unblocked_$ctr -o out --output-format csv -- $D unblocked 8192 > $O/pmc_unblocked_$ctr.log 2>&1; echo "rc=$?" >> $O/pmc_unblocked_$ctr.log )
  tail -1 $O/pmc_unblocked_$ctr.log
done
cd $R
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-residual"
DB=$(find $O/prof_blocked -name "*.db" | head -1)
python tools/prof_summary.py $DB $O/blocked32768_kernel_stats.csv "$CMD (2 factorisations in the trace)" | tail -1
python tools/prof_summary.py --by-stream $DB $O/blocked32768_kernel_stats_by_stream.csv "$CMD" | tail -1
python tools/prof_summary.py --per-launch $DB $O/blocked32768_per_launch.csv "$CMD" | tail -1; gzip -f $O/blocked32768_per_launch.csv
python tools/prof_summary.py $(find $O/prof_unblocked -name "*.db" | head -1) $O/unblocked8192_kernel_stats.csv "python bench.py --config unblocked --steps 1 --warmup 1 --no-cpu-baseline --no-residual (2 factorisations in the trace)" | tail -1
find $O -name "*.db" -delete; find $O/pmc -name "*kernel_trace.csv" -delete; find $O/pmc -name "*agent_info.csv" -delete
du -sh $O; cat $O/sizes.txt $O/bench_complex8192.txt $O/logical_ranks.txt
