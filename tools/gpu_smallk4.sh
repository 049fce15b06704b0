#!/bin/bash
# First GPU call of the next round: time the next generations of the single-workgroup panel kernels
# (DHQR_SMALLK=4: one barrier per step; 5: + blocked inverses) against the default one and run the whole
# gpu-marked suite under them.  Outputs -> gpurun_out/smallk4_*.
#   gpurun --timeout 900 -- 'bash tools/gpu_smallk4.sh'
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for k in 3 4 5; do
  DHQR_SMALLK=$k timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline \
    > gpurun_out/smallk4_bench_k$k.json 2> gpurun_out/smallk4_bench_k$k.err
  # uncontended per-phase timings (no look-ahead): the panel lane is what the kernels change
  DHQR_SMALLK=$k DHQR_LOOKAHEAD=0 timeout 200 python tools/quick_bench.py 32768,128 \
    > gpurun_out/smallk4_phases_k$k.txt 2>&1
done
for k in 4 5; do
  DHQR_SMALLK=$k timeout 500 python -m pytest tests -q -m gpu -x > gpurun_out/smallk4_pytest_k$k.txt 2>&1
  echo "pytest rc=$?" >> gpurun_out/smallk4_pytest_k$k.txt
  tail -3 gpurun_out/smallk4_pytest_k$k.txt
done
python - <<'PY'
import json
for k in (3, 4, 5):
    try:
        d = json.loads(open(f"gpurun_out/smallk4_bench_k{k}.json").read().strip().splitlines()[-1])
        print(f"SMALLK={k}: {d['value']:.0f} GFLOP/s, {d['ms_per_step']:.1f} ms/step, panel {d['phase_ms_per_step'].get('ms_panel')}")
    except Exception as e:
        print(k, "no bench line:", e)
PY
