cd /root/repo
timeout 1500 python tools/gpu_fuzz.py 31 60 2>&1 | grep -v amdgpu > gpurun_out/fuzz31.txt
tail -n 25 gpurun_out/fuzz31.txt
