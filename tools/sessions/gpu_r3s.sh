#!/bin/bash
# round 3, call s: quad steps with the step's head moved from the wide stream to the lane; ComplexF64 look-ahead with the pipelined panel kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3s; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_complex.py -x -q -k "blocked or full_size or lapack or fast_panel or complex" 2>&1 | tail -4 ) > $O/pytest_subset.txt
{
for cfg in "DHQR_QUAD=1" "DHQR_QUAD=1 DHQR_QUAD_MIN_COLS=10240" "DHQR_QUAD=1 DHQR_QUAD_MIN_COLS=3072" "DHQR_QUAD=0"; do
  echo "== $cfg"
  env $cfg python tools/quick_bench.py 32768,128 16384,128 24576,128 2>&1 | grep -v "^mfma\|amdgpu.ids" | cut -c1-420
done
} > $O/ab_quad_lane_head.txt 2>&1
{
for cfg in "DHQR_LOOKAHEAD=1" "DHQR_LOOKAHEAD=0"; do
  echo "== $cfg"
  for n in 8192 4096 16384; do env $cfg timeout 300 python tools/c64_bench.py $n 64 2>&1 | grep -v amdgpu | tail -1; done
done
} > $O/c64_lookahead.txt 2>&1
cat $O/pytest_subset.txt $O/ab_quad_lane_head.txt $O/c64_lookahead.txt | cut -c1-330
