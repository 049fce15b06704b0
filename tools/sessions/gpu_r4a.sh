#!/bin/bash
# round 4, run A: the double-double ComplexF64 solve against the plain one (ratio table), the new distributed-solve tests,
# the bench line with `also`, logical ranks with quad steps at P > 1
mkdir -p gpurun_out
python tools/c64_ratio_table.py --out gpurun_out/r04_c64_ratio_table_dd.json > gpurun_out/r4a_ratio_dd.log 2>&1
python -m pytest tests -m gpu -x -q -k "darray or complex_column_split or single_draw or acceptance_inequality_complex or device_path_vs_oracle or golden" 2>&1 | tail -40 > gpurun_out/r4a_tests.log
python bench.py --steps 5 --warmup 2 > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err
for R in 2 4; do python bench.py --logical-ranks $R --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4a_logical_$R.json 2> gpurun_out/r4a_logical_$R.err; done
DHQR_QUAD=0 python bench.py --logical-ranks 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4a_logical_2_noquad.json 2>/dev/null
