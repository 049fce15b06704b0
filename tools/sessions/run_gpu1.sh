set -x
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_complex.py -x -q -k "column_split or blocked_complex_vs_oracle" 2>&1 | tail -15 > gpurun_out/zsplit_pytest.txt
timeout 600 python tools/zsplit_probe.py > gpurun_out/zsplit_probe.txt 2>&1
tail -n 20 gpurun_out/zsplit_pytest.txt; tail -n 20 gpurun_out/zsplit_probe.txt
