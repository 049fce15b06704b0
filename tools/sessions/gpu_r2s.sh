#!/bin/bash
# session sanity run: gpu tests, contract bench, tall-skinny row-split bench
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -rfEs -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/s_pytest.txt
tail -3 gpurun_out/s_pytest.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/s_bench_blocked.json 2> gpurun_out/s_bench_blocked.err; tail -c 600 gpurun_out/s_bench_blocked.json
timeout 300 python bench.py --config tallskinny --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/s_bench_tallskinny.json 2> gpurun_out/s_bench_tallskinny.err; tail -c 900 gpurun_out/s_bench_tallskinny.json
