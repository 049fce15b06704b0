cd /root/repo
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/final_pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.txt 2>&1
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
tail -n 5 gpurun_out/final_pytest_gpu.txt gpurun_out/final_smoke.txt; head -c 1500 gpurun_out/final_bench.json
