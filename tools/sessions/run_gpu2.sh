cd /root/repo
for e in "X=1" "DHQR_SPARE_CUS=32" "DHQR_SPARE_CUS=64" "DHQR_NN_SPLIT_COLS=1" "DHQR_SPARE_CUS=64 DHQR_NN_SPLIT_COLS=1"; do
  echo "== $e"; env $e python tools/zbench.py 8192 2>&1 | grep -v amdgpu.ids
done > gpurun_out/zbench_knobs.txt 2>&1
cat gpurun_out/zbench_knobs.txt
