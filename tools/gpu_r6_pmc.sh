#!/bin/bash
# round 6: HBM counter passes only (rocprofv3 --pmc, one counter per pass, torch-free driver) + the stamp for bench.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6pmc; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
hipcc -O2 -std=c++17 $R/tools/pmc_driver.cpp -o $R/tools/pmc_driver -L $R/distributedhouseholderqr.jl_amd -ldhqr_bench -Wl,-rpath,'$ORIGIN/../distributedhouseholderqr.jl_amd' > $O/pmc_driver_build.log 2>&1
D=$R/tools/pmc_driver
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( timeout 500 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc/blocked_$ctr -o out --output-format csv -- $D blocked 32768 > $O/pmc_blocked_$ctr.log 2>&1; echo "rc=$?" >> $O/pmc_blocked_$ctr.log )
  tail -1 $O/pmc_blocked_$ctr.log
  ( timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc/unblocked_$ctr -o out --output-format csv -- $D unblocked 8192 > $O/pmc_unblocked_$ctr.log 2>&1; echo "rc=$?" >> $O/pmc_unblocked_$ctr.log )
  tail -1 $O/pmc_unblocked_$ctr.log
done
cd $R
find $O/pmc -name "*kernel_trace.csv" -delete; find $O/pmc -name "*agent_info.csv" -delete
python tools/pmc_summary.py $O/pmc $O/pmc_summary.json > $O/pmc_summary.txt 2>&1; python tools/pmc_stamp.py $O/pmc 32768 "round 6 final tree" > $O/pmc_stamp.txt 2>&1; cp profiles/pmc_traffic_current.json $O/pmc_traffic_current.json
tail -4 $O/pmc_stamp.txt
timeout 600 python bench.py --no-also > $O/bench_noalso.json 2> $O/bench.err; tail -c 700 $O/bench_noalso.json
