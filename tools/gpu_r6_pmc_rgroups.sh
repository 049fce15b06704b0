#!/bin/bash
# round 6: FETCH_SIZE of the blocked 32768^2 factorisation with k_gemm_tn2's row groups off / by height / 8 (DHQR_TUNE tn2_rgroups)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6rg; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
hipcc -O2 -std=c++17 $R/tools/pmc_driver.cpp -o $R/tools/pmc_driver -L $R/distributedhouseholderqr.jl_amd -ldhqr_bench -Wl,-rpath,'$ORIGIN/../distributedhouseholderqr.jl_amd' > $O/pmc_driver_build.log 2>&1
D=$R/tools/pmc_driver
for rg in 1 0 8; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    ( DHQR_TUNE="tn2_rgroups=$rg" timeout 500 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc_rg$rg/blocked_$ctr -o out --output-format csv -- $D blocked 32768 > $O/pmc_rg${rg}_$ctr.log 2>&1; echo "rc=$?" >> $O/pmc_rg${rg}_$ctr.log )
    tail -1 $O/pmc_rg${rg}_$ctr.log
  done
  find $O/pmc_rg$rg -name "*kernel_trace.csv" -delete; find $O/pmc_rg$rg -name "*agent_info.csv" -delete
  cd $R; python tools/pmc_summary.py $O/pmc_rg$rg $O/pmc_summary_rg$rg.json > $O/pmc_summary_rg$rg.txt 2>&1; cd /tmp
  echo "== tn2_rgroups=$rg"; grep -i "tn2\|reduce_pieces\|nn_quad" $O/pmc_summary_rg$rg.txt | head -8
done
find $O -name "*.csv" -size +2M -delete
