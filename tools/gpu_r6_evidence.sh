#!/bin/bash
# round 6 evidence on the GPU box: gpu tests, smoke, the contract bench line (with `also` + host_in_out_ms), rocprofv3
# kernel-trace summaries of the same command, HBM counter passes (one counter per pass) -> gpurun_out/r6e2/
# usage: gpu_r6_evidence.sh [skip-tests]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e2; mkdir -p $O; cd $R
if [ "$1" != "skip-tests" ]; then
( timeout 1700 python -m pytest tests -m gpu -q --timeout 900 -rfEs -p no:cacheprovider 2>&1 | tail -15 ) > $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
fi
( timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -4 ) > $O/smoke.txt; cat $O/smoke.txt
timeout 1200 python bench.py > $O/bench_blocked32768.json 2> $O/bench_blocked32768.err; tail -c 400 $O/bench_blocked32768.json
( for nb in 64 0; do timeout 300 python tools/c64_bench.py 8192 $nb 2>&1 | grep -v amdgpu | tail -1; done ) > $O/bench_complex8192.txt
( for LR in 1 2 4 8; do timeout 300 python bench.py --logical-ranks $LR --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('logical ranks', $LR, 'ms', d['ms_per_step'], 'GFLOP/s', d['value'], 'resid', d.get('residual'), 'per-rank', d.get('per_rank_ms_per_step'))"; done ) > $O/logical_ranks.txt
( timeout 600 python tools/quick_bench.py 8192,128 16384,128 24576,128 12288,0 16384,0 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['m'],'x',d['n'],'nb',d['nb'],'ms', round(d['t1']*1e3,2), 'GFLOP/s', round(d['gflops'],1), 'resid', d.get('resid'))" ) > $O/sizes.txt
timeout 300 python tools/hostio_bench.py 32768 2 > $O/hostio.txt 2>&1
( timeout 300 python tools/solve_bench.py 2>&1 | grep "^{"; DHQR_KEEP_T=0 timeout 300 python tools/solve_bench.py 8192,8192 32768,32768 2>&1 | grep "^{"; DHQR_SOLVE_PIPE=2 timeout 300 python tools/solve_bench.py 8192,8192 32768,32768 2>&1 | grep "^{"; DHQR_SOLVE_PIPE=0 timeout 300 python tools/solve_bench.py 4400,4000 8192,8192 2>&1 | grep "^{" ) > $O/solve.txt
( for i in 1 2 3; do timeout 200 python tools/small_bench.py 30 2>&1 | grep "^{"; done; echo "== DHQR_SMALL=0"; DHQR_SMALL=0 timeout 200 python tools/small_bench.py 30 2>&1 | grep "^{" ) > $O/small_bench.txt
( timeout 400 python tools/mg_solve_bench.py 2 8 2>&1 | grep "^{"; DHQR_SOLVE_PIPE=0 timeout 400 python tools/mg_solve_bench.py 2 8 2>&1 | grep "^{" ) > $O/mg_solve.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_blocked -o blocked -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-residual --no-also > $O/prof_blocked.log 2>&1
hipcc -O2 -std=c++17 $R/tools/pmc_driver.cpp -o $R/tools/pmc_driver -L $R/distributedhouseholderqr.jl_amd -ldhqr_bench -Wl,-rpath,'$ORIGIN/../distributedhouseholderqr.jl_amd' > $O/pmc_driver_build.log 2>&1
D=$R/tools/pmc_driver
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( timeout 500 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc/blocked_$ctr -o out --output-format csv -- $D blocked 32768 > $O/pmc_blocked_$ctr.log 2>&1; echo "rc=$?" >> $O/pmc_blocked_$ctr.log )
  tail -1 $O/pmc_blocked_$ctr.log
  ( timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc/unblocked_$ctr -o out --output-format csv -- $D unblocked 8192 > $O/pmc_unblocked_$ctr.log 2>&1; echo "rc=$?" >> $O/pmc_unblocked_$ctr.log )
  tail -1 $O/pmc_unblocked_$ctr.log
done
cd $R
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-residual --no-also"
DB=$(find $O/prof_blocked -name "*.db" | head -1)
python tools/prof_summary.py $DB $O/blocked32768_kernel_stats.csv "$CMD (2 factorisations in the trace)" | tail -1
python tools/prof_summary.py --by-stream $DB $O/blocked32768_kernel_stats_by_stream.csv "$CMD" | tail -1
python tools/prof_summary.py --per-launch $DB $O/blocked32768_per_launch.csv "$CMD" | tail -1; gzip -f $O/blocked32768_per_launch.csv
python tools/lane_gaps.py $O/blocked32768_per_launch.csv.gz --around 10 > $O/lane_gaps.txt 2>&1
find $O -name "*.db" -delete; find $O/pmc -name "*kernel_trace.csv" -delete; find $O/pmc -name "*agent_info.csv" -delete
python tools/pmc_summary.py $O/pmc $O/pmc_summary.json > $O/pmc_summary.txt 2>&1; python tools/pmc_stamp.py $O/pmc 32768 "round 6 final tree" > $O/pmc_stamp.txt 2>&1; cp profiles/pmc_traffic_current.json $O/pmc_traffic_current.json 2>/dev/null
du -sh $O; cat $O/sizes.txt $O/bench_complex8192.txt $O/logical_ranks.txt $O/small_bench.txt
