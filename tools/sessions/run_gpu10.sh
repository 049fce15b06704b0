cd /root/repo
timeout 600 python bench.py --logical-ranks 2 --no-cpu-baseline > gpurun_out/final_bench_logical2.json 2>gpurun_out/final_bench_logical2.err
timeout 600 python bench.py --logical-ranks 8 --no-cpu-baseline > gpurun_out/final_bench_logical8.json 2>>gpurun_out/final_bench_logical2.err
head -c 900 gpurun_out/final_bench_logical2.json; echo; head -c 900 gpurun_out/final_bench_logical8.json; tail -n 3 gpurun_out/final_bench_logical2.err
