#!/bin/bash
# round 3, call c: persistent wide GEMMs A/B (k_gemm_nn2 vs k_gemm_nn_sub, spare CUs), kernels alone, parity subset
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "blocked_vs_oracle or full_size_properties or logical_ranks or lapack" 2>&1 | tail -6 ) > $O/pytest_subset.txt
{
for cfg in "DHQR_NN2=0" "DHQR_NN2=1" "DHQR_NN2=1 DHQR_SPARE_CUS=8" "DHQR_NN2=1 DHQR_SPARE_CUS=16" "DHQR_NN2=0 DHQR_SPARE_CUS=8" "DHQR_NN2=1 DHQR_SPARE_CUS=32"; do
  echo "== $cfg"
  env $cfg python tools/quick_bench.py 32768,128 16384,128 8192,128 2>&1 | grep -v "^mfma" | cut -c1-900
done
} > $O/ab_persistent.txt 2>&1
{
for cfg in "DHQR_NN2=0" "DHQR_NN2=1" "DHQR_NN2=1 DHQR_SPARE_CUS=8"; do
  echo "== $cfg"; env $cfg python tools/gemm_bench.py 0 32768 32768 20 1 32768 32768 20 0 16384 16384 40 1 16384 16384 40 2>&1 | tail -6
done
} > $O/gemm_alone.txt 2>&1
cat $O/pytest_subset.txt; cat $O/ab_persistent.txt | cut -c1-400; cat $O/gemm_alone.txt
