#!/bin/bash
# evidence for the K-reflectors-per-pass unblocked path (BASELINE configs[1]): HBM counters, kernel stats, bench line
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc8; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D=$R/tools/pmc_driver
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( timeout 500 rocprofv3 --pmc $ctr --kernel-trace -d $O/unblocked_$ctr -o out --output-format csv -- $D unblocked 8192 > $O/unblocked_$ctr.log 2>&1; echo "rc=$?" >> $O/unblocked_$ctr.log )
  tail -1 $O/unblocked_$ctr.log
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc8 gpurun_out/pmc8/summary.json 2>&1 | head -30
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_unb -o q -- python $R/bench.py --config unblocked --steps 1 --warmup 1 --no-cpu-baseline --no-residual > $R/gpurun_out/prof_unb.log 2>&1
cd $R
python tools/prof_summary.py $(find gpurun_out/prof_unb -name "*.db" | head -1) gpurun_out/r3d_unblocked8192_kernel_stats.csv "python bench.py --config unblocked --steps 1 --warmup 1 --no-cpu-baseline --no-residual (2 factorisations in the trace)" | tail -1
find gpurun_out -name "*.db" -delete
timeout 600 python bench.py --config unblocked --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r3d_bench_unblocked8192.json
cut -c1-1500 gpurun_out/r3d_bench_unblocked8192.json
