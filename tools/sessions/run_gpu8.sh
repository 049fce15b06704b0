cd /root/repo
timeout 600 python bench.py --config unblocked --no-cpu-baseline > gpurun_out/final_bench_unblocked.json 2>/dev/null
timeout 600 python bench.py --config tallskinny --no-cpu-baseline > gpurun_out/final_bench_tallskinny.json 2>/dev/null
timeout 600 python bench.py --gpus 2 --logical-ranks 2 --no-cpu-baseline > gpurun_out/final_bench_logical2.json 2>gpurun_out/final_bench_logical2.err
python tools/zbench.py 8192 2>&1 | grep -v amdgpu > gpurun_out/final_zbench.txt
python tools/zbench.py 16384 2>&1 | grep -v amdgpu >> gpurun_out/final_zbench.txt
for f in gpurun_out/final_bench_unblocked.json gpurun_out/final_bench_tallskinny.json gpurun_out/final_bench_logical2.json; do head -c 420 $f; echo; done; cat gpurun_out/final_zbench.txt
