#!/bin/bash
# r4: more than 5 reflectors per pass where the CU can hold them (DHQR_RANKK_MAX=8, default) against 5 everywhere (=5)
mkdir -p gpurun_out
{
for KM in 5 8 5 8; do
  DHQR_RANKK_MAX=$KM python bench.py --config unblocked --steps 3 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('rankk_max $KM 8192^2', d['ms_per_step'], d['value'], d['residual'])"
done
for S in "4096 4096" "6144 6144" "8192 2048" "12288 12288"; do set -- $S; for KM in 5 8; do
  DHQR_RANKK_MAX=$KM python bench.py --config unblocked --m $1 --n $2 --steps 2 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('rankk_max $KM $1 x $2', d['ms_per_step'], d['value'], d['residual'])"
done; done
} > gpurun_out/r4ac_rankk_max.txt 2>&1
cat gpurun_out/r4ac_rankk_max.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unblocked" > gpurun_out/r4ac_pytest.txt 2>&1; tail -3 gpurun_out/r4ac_pytest.txt
