#!/bin/bash
# per-launch profile of the unblocked path (k_rankk_fused) at 8192^2 for K = 2, 3
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for K in 5; do
  DHQR_RANKK=$K timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_rk$K -o q -- python $R/tools/quick_bench.py 8192,0 > $R/gpurun_out/prof_rk$K.log 2>&1
  python $R/tools/rankk_profile.py $(find $R/gpurun_out/prof_rk$K -name "*.db" | head -1) 8192 $K 2 | tee $R/gpurun_out/r3b_rankk_profile_K$K.txt | tail -40
done
find $R/gpurun_out -name "*.db" -delete
