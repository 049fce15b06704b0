#!/bin/bash
# HBM byte counters of the two bench configurations (torch-free driver, one counter per pass)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D=$R/tools/pmc_driver
for cfg in "blocked 32768" "unblocked 8192"; do
  set -- $cfg
  for ctr in FETCH_SIZE WRITE_SIZE; do
    ( timeout 400 rocprofv3 --pmc $ctr --kernel-trace -d $O/$1_$ctr -o out --output-format csv -- $D $1 $2 > $O/$1_$ctr.log 2>&1; echo "rc=$?" >> $O/$1_$ctr.log )
    tail -1 $O/$1_$ctr.log
  done
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc3 > gpurun_out/pmc3/summary.txt 2>&1; cat gpurun_out/pmc3/summary.txt | head -60
find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*.csv" -size +30M -exec gzip {} \;
du -sh $O
