#!/bin/bash
# fused + trimmed k_panel_top: parity subset, size sweep with / without look-ahead, kernel durations
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/w_pytest.txt
tail -3 gpurun_out/w_pytest.txt
bash tools/gpu_r2v.sh > /dev/null 2>&1; cp gpurun_out/v_la_sizes.txt gpurun_out/w_la_sizes.txt; cat gpurun_out/w_la_sizes.txt
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('32768: ms', d['ms_per_step'], 'GFLOP/s', d['value'], d['phase_ms_per_step'])" | tee gpurun_out/w_bench.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_w -o w -- python $R/tools/quick_bench.py 8192,128 > $R/gpurun_out/prof_w.log 2>&1
cd $R
python tools/prof_summary.py $(find gpurun_out/prof_w -name "*.db" | head -1) gpurun_out/w_quick8192_kernel_stats.csv "python tools/quick_bench.py 8192,128" | tail -1
grep -E "k_panel_top|k_build_t" gpurun_out/w_quick8192_kernel_stats.csv
find gpurun_out -name "*.db" -size +20M -delete
