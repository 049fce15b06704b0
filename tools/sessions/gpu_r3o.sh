#!/bin/bash
# round 3, call o: quad steps (two pairs in one K = 512 update) A/B on the blocked factorisation + parity subset
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3o; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "blocked_vs_oracle or full_size_properties or lapack or fast_panel" 2>&1 | tail -6 ) > $O/pytest_subset.txt
{
for cfg in "DHQR_QUAD=0" "DHQR_QUAD=1" "DHQR_QUAD=1 DHQR_QUAD_MIN_COLS=3072" "DHQR_QUAD=1 DHQR_QUAD_MIN_COLS=10240" "DHQR_QUAD=1 DHQR_QUAD_MIN_COLS=16384"; do
  echo "== $cfg"
  env $cfg python tools/quick_bench.py 32768,128 16384,128 8192,128 2>&1 | grep -v "^mfma\|amdgpu.ids" | cut -c1-700
done
} > $O/ab_quad.txt 2>&1
cat $O/pytest_subset.txt; cat $O/ab_quad.txt | cut -c1-330
