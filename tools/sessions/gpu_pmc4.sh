#!/bin/bash
# HBM bytes of the two wide kernels ALONE (no lane kernels running beside them) on a 32768 x 32768 operand
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D=$R/tools/pmc_driver
for kind in 0 1; do for ctr in FETCH_SIZE WRITE_SIZE; do
  ( timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $O/gemm${kind}_$ctr -o out --output-format csv -- $D gemm $kind 32768 32768 3 > $O/gemm${kind}_$ctr.log 2>&1; echo "rc=$?" >> $O/gemm${kind}_$ctr.log )
  tail -2 $O/gemm${kind}_$ctr.log | head -1
done; done
cd $R
python tools/pmc_summary.py gpurun_out/pmc5 gpurun_out/pmc5/summary.json 2>&1 | head -30
