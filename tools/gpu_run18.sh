#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -rfEs -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/final_pytest.txt
timeout 600 python bench.py --config tallskinny --steps 3 --warmup 1 2>/dev/null | tail -1 | tee gpurun_out/final_bench_tallskinny.json | cut -c1-900
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/final_bench_blocked.json; python -c "
import json; d=json.loads(open('gpurun_out/final_bench_blocked.json').read()); print(d['value'], d['ms_per_step'], d['residual'], d['roofline']['frac'], d['panels_fast_fallback'])"
