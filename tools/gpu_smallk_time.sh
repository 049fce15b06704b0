#!/bin/bash
# Time the three generations of the single-workgroup panel kernels (bench + uncontended phases).
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for k in 3 4 5; do
  DHQR_SMALLK=$k timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-residual \
    > gpurun_out/smallk_bench_k$k.json 2> gpurun_out/smallk_bench_k$k.err
  DHQR_SMALLK=$k DHQR_LOOKAHEAD=0 timeout 200 python tools/quick_bench.py 32768,128 \
    > gpurun_out/smallk_phases_k$k.txt 2>&1
done
python - <<'PY'
import json
for k in (3, 4, 5):
    try:
        d = json.loads(open(f"gpurun_out/smallk_bench_k{k}.json").read().strip().splitlines()[-1])
        print(f"SMALLK={k}: {d['value']:.0f} GFLOP/s, {d['ms_per_step']:.1f} ms/step, phases {d['phase_ms_per_step']}")
    except Exception as e:
        print(k, "no bench line:", e)
    try:
        print(open(f"gpurun_out/smallk_phases_k{k}.txt").read().strip().splitlines()[-1][:600])
    except Exception as e:
        print(e)
PY
