cd /root/repo
timeout 900 python -m pytest tests/test_gpu_complex.py -x -q -k "darray or column_split" 2>&1 | tail -4 > gpurun_out/z4.txt
for i in 1 2 3 4 5; do timeout 300 python -m pytest tests/test_gpu_complex.py -x -q -k "column_split and (8-1500 or 3-700 or 3-9000)" 2>&1 | tail -1 >> gpurun_out/z4.txt; done
cat gpurun_out/z4.txt
