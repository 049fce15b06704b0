#!/bin/bash
# HBM byte counters of the 32768^2 blocked factorisation, final kernels of the round (torch-free driver, one counter per pass)
# build first (cross-compiles without a GPU):
#   hipcc -O2 -std=c++17 tools/pmc_driver.cpp -o tools/pmc_driver -L distributedhouseholderqr.jl_amd -ldhqr_bench -Wl,-rpath,'$ORIGIN/../distributedhouseholderqr.jl_amd'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D=$R/tools/pmc_driver
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( timeout 500 rocprofv3 --pmc $ctr --kernel-trace -d $O/blocked_$ctr -o out --output-format csv -- $D blocked 32768 > $O/blocked_$ctr.log 2>&1; echo "rc=$?" >> $O/blocked_$ctr.log )
  tail -1 $O/blocked_$ctr.log
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc4 gpurun_out/pmc4/summary.json > gpurun_out/pmc4/summary.txt 2>&1; head -30 gpurun_out/pmc4/summary.txt
find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*.csv" -size +30M -exec gzip {} \;
du -sh $O
