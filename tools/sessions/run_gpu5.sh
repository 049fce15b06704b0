cd /root/repo
timeout 900 python -m pytest tests/test_gpu_complex.py -x -q -k "darray or column_split" 2>&1 | tail -4 > gpurun_out/z5.txt
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_complex.py -x -q -k "column_split and (8-1500 or 3-700 or 3-9000)" 2>&1 | tail -1 >> gpurun_out/z5.txt; done
timeout 600 python tools/zsplit_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/z5.txt
cat gpurun_out/z5.txt
