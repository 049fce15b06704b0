#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q; mkdir -p $O; cd $R
{
for cfg in "DHQR_ZPIPE=0" "DHQR_ZPIPE=1" "DHQR_ZPIPE=3"; do
  echo "== $cfg"
  for n in 8192 4096 2048 1024; do env $cfg timeout 300 python tools/c64_bench.py $n 64 2>&1 | grep -v amdgpu | tail -1; done
done
} > $O/c64_ab4.txt 2>&1
cat $O/c64_ab4.txt
