#!/bin/bash
# r4: seven reflectors per pass at 6145 ... 8192 rows (two column buffers, five reflectors in registers) against six
mkdir -p gpurun_out
{
for KM in 6 8 6 8; do
  DHQR_RANKK_MAX=$KM python bench.py --config unblocked --steps 3 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('rankk_max $KM 8192^2', d['ms_per_step'], d['value'], d['residual'])"
done
for S in "6144 6144" "12288 12288" "8192 6000"; do set -- $S; for KM in 6 8; do
  DHQR_RANKK_MAX=$KM python bench.py --config unblocked --m $1 --n $2 --steps 2 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('rankk_max $KM $1 x $2', d['ms_per_step'], d['value'], d['residual'])"
done; done
} > gpurun_out/r4ah_rankk7.txt 2>&1
cat gpurun_out/r4ah_rankk7.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unblocked" > gpurun_out/r4ah_pytest.txt 2>&1; tail -3 gpurun_out/r4ah_pytest.txt
