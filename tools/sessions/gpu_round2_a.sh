#!/bin/bash
# round 2, first check of the rewritten driver: GPU test-suite, single-GPU bench, logical ranks on one GPU
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/r2a_pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a_pytest.txt
tail -15 gpurun_out/r2a_pytest.txt
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2a_bench1.json 2> gpurun_out/r2a_bench1.err
tail -c 1500 gpurun_out/r2a_bench1.json; tail -3 gpurun_out/r2a_bench1.err
for r in 1 2 4 8; do
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --logical-ranks $r > gpurun_out/r2a_bench_lr$r.json 2> gpurun_out/r2a_bench_lr$r.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r2a_bench_lr$r.json").read().strip().splitlines()[-1])
    print("logical ranks $r:", round(d["value"]), "GFLOP/s", round(d["ms_per_step"],1), "ms  resid", d["residual"], d["panels_fast_fallback"], d.get("per_rank_gemm_ms_per_step"), d.get("per_rank_panel_ms_per_step"))
except Exception as e:
    print("lr $r failed", e); print(open("gpurun_out/r2a_bench_lr$r.err").read()[-1500:])
PY
done
