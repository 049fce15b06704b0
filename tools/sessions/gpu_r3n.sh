#!/bin/bash
# round 3, call n: NN with K = 512 (k_gemm_nn_sub<2,512>) vs K = 256 alone; non-temporal C accesses
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3n; mkdir -p $O; cd $R
{
for cfg in "DHQR_NN2=0" "DHQR_NN2=0 DHQR_NTC=1"; do
  echo "== $cfg"; env $cfg python tools/gemm_bench.py 0 32768 32768 20 2 32768 32768 10 0 16384 16384 40 2 16384 16384 20 0 32768 8192 40 2 32768 8192 20 2>&1 | grep -v amdgpu.ids
done
} > $O/nn512.txt 2>&1
cat $O/nn512.txt
