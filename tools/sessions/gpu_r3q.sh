#!/bin/bash
# round 3, call q: ComplexF64 column-pipelined panel kernel (k_zpanel_pipe): tests, timing A/B, repeatability
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q; mkdir -p $O; cd $R
( timeout 1200 python -m pytest tests/test_gpu_complex.py -x -q 2>&1 | tail -6 ) > $O/pytest_complex.txt
{
for cfg in "DHQR_ZPIPE=0" "DHQR_ZPIPE=1"; do
  echo "== $cfg"
  for n in 8192 4096 16384; do env $cfg timeout 300 python tools/c64_bench.py $n 64 2>&1 | grep -v amdgpu | tail -1; done
done
} > $O/c64_ab.txt 2>&1
cat $O/pytest_complex.txt $O/c64_ab.txt
