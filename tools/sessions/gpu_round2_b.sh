#!/bin/bash
# A/B of the panel-chain kernels: 1024-thread (5) vs 256-thread fused (6): bench, uncontended phases, kernel trace
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for k in 5 6; do
  DHQR_SMALLK=$k timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-residual > gpurun_out/r2b_bench_k$k.json 2> gpurun_out/r2b_bench_k$k.err
  DHQR_SMALLK=$k DHQR_LOOKAHEAD=0 timeout 200 python tools/quick_bench.py 32768,128 > gpurun_out/r2b_phases_k$k.txt 2>&1
  DHQR_SMALLK=$k timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-residual --logical-ranks 8 > gpurun_out/r2b_lr8_k$k.json 2> gpurun_out/r2b_lr8_k$k.err
done
DHQR_SMALLK=6 timeout 600 python -m pytest tests -q -m gpu -x -k "parity or kernels" > gpurun_out/r2b_pytest_k6.txt 2>&1; echo "rc=$?" >> gpurun_out/r2b_pytest_k6.txt; tail -3 gpurun_out/r2b_pytest_k6.txt
bash tools/gpu_prof.sh r2b_k6 DHQR_SMALLK=6 > gpurun_out/r2b_prof.log 2>&1
python - <<'PY'
import json
for k in (5, 6):
    for tag in ("bench", "lr8"):
        try:
            d = json.loads(open(f"gpurun_out/r2b_{tag}_k{k}.json").read().strip().splitlines()[-1])
            print(f"SMALLK={k} {tag}: {d['value']:.0f} GFLOP/s, {d['ms_per_step']:.1f} ms/step, phases {d['phase_ms_per_step']}")
        except Exception as e:
            print(k, tag, "no bench line:", e)
    print(open(f"gpurun_out/r2b_phases_k{k}.txt").read().strip().splitlines()[-1][:400])
PY
head -30 gpurun_out/prof_r2b_k6_kernel_stats.csv | cut -c1-160
