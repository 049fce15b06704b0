#!/bin/bash
# full GPU suite (new row-split driver, full-size LAPACK/oracle pins), tall-skinny bench at 1 and 8 logical ranks
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -q -m gpu -x -s > gpurun_out/r2d_pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r2d_pytest.txt
grep -E "n=|8192\^2|passed|failed|rc=" gpurun_out/r2d_pytest.txt | tail -12
timeout 300 python bench.py --config tallskinny --steps 3 --warmup 1 > gpurun_out/r2d_ts1.json 2> gpurun_out/r2d_ts1.err; tail -c 900 gpurun_out/r2d_ts1.json; tail -2 gpurun_out/r2d_ts1.err
timeout 300 python bench.py --config tallskinny --steps 2 --warmup 1 --logical-ranks 8 > gpurun_out/r2d_ts8.json 2> gpurun_out/r2d_ts8.err; tail -c 600 gpurun_out/r2d_ts8.json; tail -2 gpurun_out/r2d_ts8.err
