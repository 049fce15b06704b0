#!/bin/bash
# round 4, run B: Float64 acceptance-statistic table, the full GPU suite on the current tree
mkdir -p gpurun_out
python tools/c64_ratio_table.py --real --out gpurun_out/r04_f64_ratio_table.json > gpurun_out/r4b_ratio_real.log 2>&1
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r4b_full.log
